"""Status codes of include/cairo_zstd_amd_status.h, parsed from the header so that the Python
names can never drift from the C ABI."""
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "cairo_zstd_amd_status.h")

CODES: dict[str, int] = {}
for _m in re.finditer(r"^\s*(CZ_(?:OK|E_[A-Z0-9_]+))\s*=\s*(\d+)", open(_HDR).read(), re.M):
    CODES[_m.group(1)] = int(_m.group(2))
NAMES = {v: k for k, v in CODES.items()}
globals().update(CODES)


def name(code: int) -> str:
    return NAMES.get(int(code), f"CZ_?{code}")


class CzError(RuntimeError):
    def __init__(self, code: int, what: str = ""):
        self.code = int(code)
        super().__init__(f"{name(code)} ({code}) {what}".strip())
