// Micro-benchmark: how a 128 B record line that a kernel only writes reaches HBM on gfx950.
//   mode 0: every lane writes its own line, eight 16 B stores one after the other (what the wavefront kernels do with a vertex' records)
//   mode 1: the same, but only seven of the eight (112 of 128 B: a line left partly unwritten)
//   mode 2: eight lanes write one line together (one 128 B request per line), the lines of a wave's 64 vertices in eight such instructions
//   mode 3: a plain coalesced stream (lane i writes element i)
//   mode 4: seven 16 B stores and three of 4 B (116 of 128 B: a POOL child vertex' first line as wf_scatter writes it)
//   mode 5: seven 16 B stores and four of 4 B (the whole line, its tail in words)
//   mode 6: the line's first half in one launch, its second half in the next (a line completed much later)
//   mode 7: mode 2 with the transposition it needs in a kernel: every lane puts its eight records into LDS first
// A second argument thins the lanes out (percent of lanes that write: the queues' waves are not full of vertices that scatter).
// hipcc --offload-arch=gfx950 -O3 tools/micro/line_writes.hip -o gpurun_out/line_writes && rocprofv3 --pmc FETCH_SIZE WRITE_SIZE ... -- gpurun_out/line_writes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ inline bool writes(size_t i, unsigned percent) {
    unsigned h = unsigned(i) * 2654435761u;
    h ^= h >> 15;
    return (h % 100u) < percent;
}

__global__ void __launch_bounds__(256) write_lines(float4* lines, size_t n, int mode, unsigned stride_lines, unsigned percent, int half) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = make_float4(float(i), 1.f, 2.f, 3.f);
    const bool   active = writes(i, percent);
    if (4 == mode || 5 == mode) {
        if (!active) return;
        float4* p = lines + ((i * stride_lines) % n) * 8;
        for (int k = 0; k < 7; ++k) p[k] = v;
        float* w = reinterpret_cast<float*>(p + 7);
        for (int k = 0; k < (4 == mode ? 3 : 4); ++k) w[k] = v.x;
        return;
    }
    if (6 == mode) {
        if (!active) return;
        float4* p = lines + ((i * stride_lines) % n) * 8 + 4 * half;
        for (int k = 0; k < 4; ++k) p[k] = v;
        return;
    }
    if (7 == mode) {
        __shared__ float4 rows[256 * 9];
        const unsigned lane = threadIdx.x & 63u;
        float4*        mine = rows + (threadIdx.x >> 6) * 64 * 9;
        for (int k = 0; k < 8; ++k) mine[lane * 9 + k] = make_float4(v.x, float(k), v.z, v.w);
        __builtin_amdgcn_wave_barrier();
        const size_t             wave_first = i - lane;
        const unsigned long long mask = __ballot(active);
        for (int j = 0; j < 8; ++j) {
            const unsigned src = 8u * j + (lane >> 3);
            if (!((mask >> src) & 1ull)) continue;
            const size_t owner = wave_first + src;
            float4*      p = lines + ((owner * stride_lines) % n) * 8;
            p[lane & 7u] = mine[src * 9 + (lane & 7u)];
        }
        return;
    }
    if (mode < 2 && !active) return;
    if (0 == mode || 1 == mode) {
        // lane i owns line (i * stride_lines) % n: scattered like the pool's vertex blocks
        float4* p = lines + ((i * stride_lines) % n) * 8;
        const int count = 0 == mode ? 8 : 7;
        for (int k = 0; k < count; ++k) p[k] = v;
    } else if (2 == mode) {
        const unsigned lane = threadIdx.x & 63u;
        const size_t   wave_first = i - lane;
        for (int j = 0; j < 8; ++j) {
            const size_t owner = wave_first + 8u * j + (lane >> 3);  // the vertex whose line these eight lanes write
            if (!writes(owner, percent)) continue;
            float4*      p = lines + ((owner * stride_lines) % n) * 8;
            p[lane & 7u] = v;
        }
    } else {
        for (int k = 0; k < 8; ++k) lines[size_t(k) * n + i] = v;
    }
}

int main(int argc, char** argv) {
    const size_t n = size_t(1) << 24;  // 16 M lines = 2 GiB
    float4*      d = nullptr;
    if (hipSuccess != hipMalloc(&d, n * 128)) return 1;
    hipMemset(d, 0, n * 128);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned stride = argc > 1 ? unsigned(std::atoi(argv[1])) : 1u;
    const unsigned percent = argc > 2 ? unsigned(std::atoi(argv[2])) : 100u;
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(write_lines, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, d, n, mode, stride, percent, 0);
            if (6 == mode) hipLaunchKernelGGL(write_lines, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, d, n, mode, stride, percent, 1);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0.f;
            hipEventElapsedTime(&ms, a, b);
            const double bytes = double(n) * (percent / 100.0) * (1 == mode ? 112 : 4 == mode ? 124 : 128);
            if (2 == rep) std::printf("mode %d stride %u, %u %% of the lanes: %.3f ms, %.2f TB/s of bytes written\n", mode, stride, percent, ms, bytes / (ms * 1e-3) / 1e12);
        }
    }
    hipFree(d);
    return 0;
}
