// Diagnostic microbenchmark: huff0 with ONE LANE PER STREAM — the cost model of a kernel for batches that have
// tens of thousands of independent streams (config 3: 40 000), before any such kernel is built.
//   hipcc --offload-arch=gfx950 -O3 huf_lane.hip -o huf_lane && ./huf_lane
// A workgroup of 256 threads decodes 256 streams = 64 blocks x 4 streams; every block has its own decoding table in LDS.  TBITS = index
// bits of the table (entries: symbol | bits << 8, 2 bytes): 11 = the full table of the format's longest code (4 KiB per block: 40 blocks
// per CU, so ONE such workgroup of 160 threads would be all a CU holds — modelled here as 64 blocks x 4 KiB = does not fit; the variant
// runs with 32 blocks per workgroup, 128 threads), 9 = a first-level table (1 KiB per block; codes longer than 9 bits would take a
// second lookup, which the symbols of this benchmark — a 5-bit code — never need; the branch that tests for it is in the loop).
// Per lane: a 64-bit window refilled 32 bits at a time from its own stream (8-byte loads, one ahead), four symbols per group packed
// into one dword store to the lane's own output.  Reported: ns per symbol-step of a wave (64 symbols), the chip's symbols per second,
// and what 1.31 G symbols (config 3) would take.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int TBITS, int BLOCKS>
__global__ void __launch_bounds__(BLOCKS * 4) k_huf(const uint32_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t words_per_stream, uint32_t nsym, uint32_t* sink) {
    extern __shared__ uint16_t tab[];                                   // BLOCKS tables of 1 << TBITS entries
    const uint32_t t = threadIdx.x, blk = t >> 2;
    // every block its own table: a 5-bit code, symbol = code ^ (block's salt), written as a real builder would (all entries of a code)
    for (uint32_t i = t; i < (uint32_t)BLOCKS << TBITS; i += blockDim.x) {
        const uint32_t b = i >> TBITS, idx = i & ((1u << TBITS) - 1u);
        tab[i] = (uint16_t)((((idx >> (TBITS - 5)) ^ b) & 31u) | (5u << 8));
    }
    __syncthreads();
    const uint16_t* my = tab + ((size_t)blk << TBITS);
    const uint64_t stream = (uint64_t)blockIdx.x * blockDim.x + t;
    const uint32_t* p = in + stream * words_per_stream;
    uint8_t* o = out + stream * nsym;
    uint64_t w = ((uint64_t)p[0] << 32) | p[1];
    uint32_t avail = 64, nxt = p[2], pos = 3, acc_all = 0;
    for (uint32_t s = 0; s < nsym; s += 4) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t e = my[(uint32_t)(w >> (64 - TBITS))];
            if (TBITS < 11 && (e & 0x8000u)) e = my[(e & 0x7FFu) + (uint32_t)((w >> (64 - 11)) & 3u)];   // second level (never taken here; the test is)
            const uint32_t nb = (e >> 8) & 15u;
            acc |= (e & 255u) << (8 * k);
            w <<= nb; avail -= nb;
        }
        *(uint32_t*)(o + s) = acc;
        acc_all ^= acc;
        if (avail <= 32) { w |= (uint64_t)nxt << (32 - avail); avail += 32; nxt = p[pos < words_per_stream ? pos : 0]; pos++; }
    }
    if (acc_all == 0x12345678u) sink[0] = acc_all;
}

template <int TBITS, int BLOCKS>
static int run(const char* what, const uint32_t* d_in, uint8_t* d_out, uint32_t* d_sink, uint32_t words, uint32_t nsym, int cus, int wg_per_cu) {
    const size_t lds = (size_t)BLOCKS << (TBITS + 1);
    CHECK(hipFuncSetAttribute((const void*)k_huf<TBITS, BLOCKS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int grid : {1, cus * wg_per_cu}) {
        hipLaunchKernelGGL((k_huf<TBITS, BLOCKS>), dim3(grid), dim3(BLOCKS * 4), lds, 0, d_in, d_out, words, 64u, d_sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_huf<TBITS, BLOCKS>), dim3(grid), dim3(BLOCKS * 4), lds, 0, d_in, d_out, words, nsym, d_sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double streams = (double)grid * BLOCKS * 4, syms = streams * nsym;
        printf("%-58s grid %5d (%6.0f streams): %8.3f ms  %7.2f ns per symbol-step of a wave  %8.1f G symbols/s  config 3 (1.31 G symbols): %6.2f ms\n",
               what, grid, streams, ms, ms * 1e6 / nsym, syms / (ms * 1e6), 1.31e9 / (syms / (ms * 1e-3)) * 1e3);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t nsym = 32768, words = nsym * 5 / 32 + 8;              // a config-3 stream: 32 768 symbols of ~5 bits
    const size_t max_streams = (size_t)cus * 4 * 256;
    uint32_t* d_in; uint8_t* d_out; uint32_t* d_sink;
    CHECK(hipMalloc(&d_in, max_streams * words * 4)); CHECK(hipMalloc(&d_out, max_streams * nsym)); CHECK(hipMalloc(&d_sink, 64));
    {
        std::vector<uint32_t> h(max_streams * words);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
        CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    printf("%d CUs; streams of %u symbols, 5 bits each; one lane per stream, four streams per block, a table per block in LDS\n", cus, nsym);
    if (run<11, 32>("full table (4 KiB per block), 32 blocks per workgroup, 1 / CU", d_in, d_out, d_sink, words, nsym, cus, 1)) return 1;
    if (run<9, 64>("9-bit first level (1 KiB per block), 64 blocks per WG, 1 / CU", d_in, d_out, d_sink, words, nsym, cus, 1)) return 1;
    if (run<9, 64>("9-bit first level (1 KiB per block), 64 blocks per WG, 2 / CU", d_in, d_out, d_sink, words, nsym, cus, 2)) return 1;
    if (run<10, 32>("10-bit first level (2 KiB per block), 32 blocks per WG, 2 / CU", d_in, d_out, d_sink, words, nsym, cus, 2)) return 1;
    return 0;
}
