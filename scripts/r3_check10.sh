#!/bin/bash
set -o pipefail
bash scripts/ktimeline.sh raw_rle 10000 rawrle || exit 1
bash scripts/ktimeline.sh mix 12500 mix || exit 1
