// tools/grad_sweep.hip -- hardware A/B harness for the HBM-bound dense-gradient writer.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/grad_sweep tools/grad_sweep.hip
// Run on the GPU box: ./tools/grad_sweep [B T U1 V]
// Every variant writes the same (B,T,U1,V) fp32 tensor; timing = hipEvents around N launches,
// variants interleaved over several rounds (cdna guide 5.4 rule 24).
#include "../pika_amd/csrc/rnnt_loss.hip"

#include <stdio.h>
#include <vector>
#include <algorithm>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

namespace {

template <bool NT, int PER_THREAD>
__global__ __launch_bounds__(256) void fill_chunk(v4f *p, size_t n4) {
    const size_t base = (size_t)blockIdx.x * (256 * PER_THREAD) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const size_t i = base + (size_t)k * 256;
        if (i < n4) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
        }
    }
}

// per-lane meta lookup (the library form), PER_THREAD as a parameter
template <bool NT, int PER_THREAD>
__global__ __launch_bounds__(256) void grad_flat_lane(const RowMeta *__restrict__ meta, size_t n4, int V4,
                                                     int blank, v4f *__restrict__ grads) {
    const int qb = blank >> 2, cb = blank & 3;
    size_t i = (size_t)blockIdx.x * (256 * PER_THREAD) + threadIdx.x;
    size_t row = i / (unsigned)V4;
    int q = (int)(i - row * V4);
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        if (i < n4) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            const int ye = meta[row].ye;
            const int qe = ye >> 2;
            if (q == qb || q == qe) {
                const float gb = meta[row].gb, ge = meta[row].ge;
                if (q == qb) { v.x = cb == 0 ? gb : 0.f; v.y = cb == 1 ? gb : 0.f; v.z = cb == 2 ? gb : 0.f; v.w = cb == 3 ? gb : 0.f; }
                if (q == qe) { const int ce = ye & 3; v.x = ce == 0 ? ge : v.x; v.y = ce == 1 ? ge : v.y; v.z = ce == 2 ? ge : v.z; v.w = ce == 3 ? ge : v.w; }
            }
            if constexpr (NT) __builtin_nontemporal_store(v, grads + i); else grads[i] = v;
        }
        i += 256; q += 256;
        while (q >= V4) { q -= V4; ++row; }
    }
}

// wave-uniform meta: a wave instruction (64 groups) spans at most two rows when V4 >= 64;
// both rows' metadata come in through the scalar cache.
template <bool NT, int PER_THREAD>
__global__ __launch_bounds__(256) void grad_flat_scalar(const RowMeta *__restrict__ meta, size_t n4, size_t nrows,
                                                       int V4, int blank, v4f *__restrict__ grads) {
    const int qb = blank >> 2, cb = blank & 3;
    const int lane = threadIdx.x & 63;
    // first group of this wave's first instruction (wave-uniform)
    size_t i0 = (size_t)blockIdx.x * (256 * PER_THREAD) + (threadIdx.x & ~63);
    i0 = ((size_t)__builtin_amdgcn_readfirstlane((int)(i0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)i0);
    size_t row0 = i0 / (unsigned)V4;
    int q0 = (int)(i0 - row0 * V4);
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const size_t i = i0 + lane;
        const size_t row1 = row0 + 1 < nrows ? row0 + 1 : row0;
        const RowMeta m0 = meta[row0], m1 = meta[row1];  // scalar loads
        int q = q0 + lane;
        const bool second = q >= V4;
        q = second ? q - V4 : q;
        const float gb = second ? m1.gb : m0.gb, ge = second ? m1.ge : m0.ge;
        const int ye = second ? m1.ye : m0.ye;
        const int qe = ye >> 2, ce = ye & 3;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (q == qb) { v.x = cb == 0 ? gb : 0.f; v.y = cb == 1 ? gb : 0.f; v.z = cb == 2 ? gb : 0.f; v.w = cb == 3 ? gb : 0.f; }
        if (q == qe) { v.x = ce == 0 ? ge : v.x; v.y = ce == 1 ? ge : v.y; v.z = ce == 2 ? ge : v.z; v.w = ce == 3 ? ge : v.w; }
        if (i < n4) { if constexpr (NT) __builtin_nontemporal_store(v, grads + i); else grads[i] = v; }
        i0 += 256; q0 += 256;
        while (q0 >= V4) { q0 -= V4; ++row0; }
    }
}

__global__ __launch_bounds__(256) void checksum_kernel(const float *p, size_t n, double *out) {
    double acc = 0.0; unsigned long long nz = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = p[i];
        if (v != 0.f) { acc += (double)v * (double)(1 + (i % 8191)); ++nz; }
    }
    atomicAdd(out, acc);
    atomicAdd(out + 1, (double)nz);
}

struct Variant { std::string name; std::function<void(hipStream_t)> run; std::vector<float> ms; };

}  // namespace

int main(int argc, char **argv) {
    int B = 32, T = 1000, U1 = 51, V = 5000;
    if (argc >= 5) { B = atoi(argv[1]); T = atoi(argv[2]); U1 = atoi(argv[3]); V = atoi(argv[4]); }
    const int blank = 0;
    const size_t nrows = (size_t)B * T * U1;
    const size_t n = nrows * V, n4 = n / 4;
    float *grads; CK(hipMalloc(&grads, n * 4));
    const size_t wsb = pika_rnnt_workspace_bytes(B, T, U1);
    void *ws; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
    std::vector<int> hl((size_t)B * (U1 - 1)), hT(B, T), hU(B, U1 - 1);
    for (size_t i = 0; i < hl.size(); ++i) hl[i] = 1 + (int)((i * 2654435761u) % (V - 1));
    int *labels, *Tn, *Un;
    CK(hipMalloc(&labels, hl.size() * 4)); CK(hipMalloc(&Tn, B * 4)); CK(hipMalloc(&Un, B * 4));
    CK(hipMemcpy(labels, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Tn, hT.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Un, hU.data(), B * 4, hipMemcpyHostToDevice));
    const Lattice L = carve(ws, B, T, U1);
    hipStream_t s; CK(hipStreamCreate(&s));
    const RowMeta *meta = L.meta;
    v4f *g4 = (v4f *)grads;

    std::vector<Variant> vs;
    vs.push_back({"hipMemsetAsync", [&](hipStream_t st) { CK(hipMemsetAsync(grads, 0, n * 4, st)); }, {}});
    vs.push_back({"fill_chunk<nt,4>", [=](hipStream_t st) { hipLaunchKernelGGL((fill_chunk<true, 4>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, st, g4, n4); }, {}});
    vs.push_back({"fill_chunk<plain,4>", [=](hipStream_t st) { hipLaunchKernelGGL((fill_chunk<false, 4>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, st, g4, n4); }, {}});
    vs.push_back({"fill_chunk<plain,2>", [=](hipStream_t st) { hipLaunchKernelGGL((fill_chunk<false, 2>), dim3((unsigned)((n4 + 511) / 512)), dim3(256), 0, st, g4, n4); }, {}});
    vs.push_back({"fill_chunk<plain,8>", [=](hipStream_t st) { hipLaunchKernelGGL((fill_chunk<false, 8>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, st, g4, n4); }, {}});
    vs.push_back({"LIB backward (rowmeta + grad)", [=](hipStream_t st) {
        int rc = pika_rnnt_loss_backward(labels, Tn, Un, B, T, U1, V, blank, nullptr, ws, grads, st);
        if (rc) { printf("lib rc=%d\n", rc); exit(1); } }, {}});
#define LANE(NTV, PT) vs.push_back({std::string("flat_lane<") + (NTV ? "nt" : "plain") + "," #PT ">", [=](hipStream_t st) { \
        hipLaunchKernelGGL((grad_flat_lane<NTV, PT>), dim3((unsigned)((n4 + 256 * PT - 1) / (256 * PT))), dim3(256), 0, st, meta, n4, V / 4, blank, g4); }, {}})
#define SCAL(NTV, PT) vs.push_back({std::string("flat_scalar<") + (NTV ? "nt" : "plain") + "," #PT ">", [=](hipStream_t st) { \
        hipLaunchKernelGGL((grad_flat_scalar<NTV, PT>), dim3((unsigned)((n4 + 256 * PT - 1) / (256 * PT))), dim3(256), 0, st, meta, n4, nrows, V / 4, blank, g4); }, {}})
    LANE(false, 1); LANE(false, 2); LANE(false, 4); LANE(false, 8); LANE(true, 4); LANE(true, 2);
    SCAL(false, 1); SCAL(false, 2); SCAL(false, 4); SCAL(false, 8); SCAL(true, 4); SCAL(true, 2);

    // correctness: every gradient variant must reproduce the library's tensor (weighted checksum)
    {
        // make the lattice non-trivial: alpha/beta planes = small pseudo-random values
        std::vector<float> h(4 * plane_elems(B, T, U1));
        for (size_t i = 0; i < h.size(); ++i) h[i] = -0.001f * (float)((i * 2654435761u) % 4096);
        CK(hipMemcpy(ws, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        double *cs; CK(hipMalloc(&cs, 16));
        auto checksum = [&](double *o) {
            CK(hipMemsetAsync(cs, 0, 16, s));
            hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, s, grads, n, cs);
            CK(hipMemcpy(o, cs, 16, hipMemcpyDeviceToHost));
        };
        double ref[2] = {0, 0};
        for (auto &v : vs) {
            if (v.name.find("fill") != std::string::npos || v.name.find("Memset") != std::string::npos) continue;
            CK(hipMemsetAsync(grads, 0xff, n * 4, s));  // poison (NaN pattern): unwritten elements show up
            v.run(s);
            double got[2]; checksum(got);
            if (v.name.find("LIB") != std::string::npos) { ref[0] = got[0]; ref[1] = got[1]; }
            printf("check %-36s sum=%.9e nnz=%.0f %s\n", v.name.c_str(), got[0], got[1],
                   (got[0] == ref[0] && got[1] == ref[1]) ? "OK" : "MISMATCH");
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 5, reps = 3;
    for (int r = 0; r < rounds; ++r) {
        for (auto &v : vs) {
            v.run(s);  // warm
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < reps; ++k) v.run(s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / reps);
        }
    }
    printf("tensor %.2f GB (B=%d T=%d U1=%d V=%d)\n", n * 4 / 1e9, B, T, U1, V);
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
        printf("%-36s median %8.3f ms  min %8.3f ms  -> %7.1f GB/s (median)\n", v.name.c_str(), med, mn, n * 4 / (med * 1e-3) / 1e9);
    }
    return 0;
}
