// Micro-benchmark (bring-up tool): how many bytes per clock can one CU pull through its vector L1 when every CU of the
// chip does the same?  Mode 0: all CUs stream the same L2-resident "weight" region (rows of 128 B, row stride given);
// mode 1: every CU streams its own region of a large buffer (HBM / MALL).  W waves per CU, D independent 16-byte loads
// per thread in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int D>
__global__ __launch_bounds__(512) void fill_kernel(const uint4* __restrict__ buf, size_t region_bytes, size_t cu_stride_bytes,
                                                   int row_stride_bytes, int iters, unsigned* sink, unsigned long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const char* base = (const char*)buf + (size_t)blockIdx.x * cu_stride_bytes;
    // a wave-load covers 8 rows x 128 B (like a weight slab piece): lane -> row lane/8, 16-byte part lane%8
    const size_t lane_off = (size_t)(lane >> 3) * row_stride_bytes + (lane & 7) * 16;
    const size_t wave_span = (size_t)8 * row_stride_bytes;       // bytes of address space one wave-load spans
    uint4 acc = make_uint4(0, 0, 0, 0);
    size_t pos = (size_t)wave * wave_span;
    const size_t step = (size_t)nw * wave_span;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            size_t o = pos + lane_off; pos += step;
            if (pos + wave_span > region_bytes) pos = (size_t)wave * wave_span;
            v[d] = *reinterpret_cast<const uint4*>(base + o);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc.x == 0x12345678u && acc.y == 17u) sink[0] = acc.z ^ acc.w;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int D>
static void run(const uint4* buf, size_t region, size_t cu_stride, int row_stride, int waves, int iters, unsigned* sink,
                unsigned long long* cyc, int ncu, const char* tag) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fill_kernel<D><<<ncu, waves * 64>>>(buf, region, cu_stride, row_stride, iters, sink, cyc);
    hipEventRecord(e0);
    fill_kernel<D><<<ncu, waves * 64>>>(buf, region, cu_stride, row_stride, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu); hipMemcpy(h.data(), cyc, ncu * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += (double)c; avg /= ncu;
    const double bytes_cu = (double)iters * D * waves * 1024.0;
    printf("%-28s waves=%d D=%2d : %7.2f B/clk/CU  (%.0f cyc/wave-load/CU)  chip %.2f TB/s  [%.3f ms]\n", tag, waves, D,
           bytes_cu / avg, avg / (iters * D * waves), bytes_cu * ncu / (ms * 1e-3) / 1e12, ms);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const size_t big = (size_t)4 << 30;
    uint4* buf; hipMalloc(&buf, big); hipMemset(buf, 1, big);
    unsigned* sink; hipMalloc(&sink, 4); unsigned long long* cyc; hipMalloc(&cyc, ncu * 8);
    printf("CUs %d\n", ncu);
    const int iters = 2000;
    for (int waves : {4, 8}) {
        // shared L2-resident weights: 295 KB region, rows 2304 B apart (Cin=128) / contiguous rows (128 B apart)
        run<2>(buf, 294912, 0, 2304, waves, iters, sink, cyc, ncu, "L2 shared stride2304");
        run<4>(buf, 294912, 0, 2304, waves, iters, sink, cyc, ncu, "L2 shared stride2304");
        run<8>(buf, 294912, 0, 2304, waves, iters, sink, cyc, ncu, "L2 shared stride2304");
        run<16>(buf, 294912, 0, 2304, waves, iters, sink, cyc, ncu, "L2 shared stride2304");
        run<8>(buf, 294912, 0, 128, waves, iters, sink, cyc, ncu, "L2 shared contiguous");
        run<16>(buf, 294912, 0, 128, waves, iters, sink, cyc, ncu, "L2 shared contiguous");
        run<8>(buf, 589824, 0, 4608, waves, iters, sink, cyc, ncu, "L2 shared stride4608");
        // private streams from HBM: each CU walks its own 16 MB
        run<4>(buf, (size_t)16 << 20, (size_t)16 << 20, 256, waves, iters / 4, sink, cyc, ncu, "HBM private stride256");
        run<8>(buf, (size_t)16 << 20, (size_t)16 << 20, 256, waves, iters / 4, sink, cyc, ncu, "HBM private stride256");
        run<16>(buf, (size_t)16 << 20, (size_t)16 << 20, 256, waves, iters / 4, sink, cyc, ncu, "HBM private stride256");
    }
    return 0;
}
