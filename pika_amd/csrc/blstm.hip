// pika_amd/csrc/blstm.hip -- one (bi)directional LSTM layer over a padded batch as ONE persistent launch for gfx950
// (include/pika_las.h: pika_blstm_*; reference trainer/model/las.py:44-75, the rescorer's nn.LSTM encoder over
// pack_padded_sequence input).
//
// The recurrence is a chain of S dependent steps of a (B x H) x (H x 4H) product: 134 MFLOP at B = 64, H = 512, i.e.
// nothing -- a library LSTM spends its time on the three launches per step and direction.  Here the recurrent weights
// never move: a workgroup owns 16 hidden units (all four gates: 64 weight rows) of one direction for 16 batch rows and
// keeps its 64 x H slice in REGISTERS for the whole sequence, as two bf16 terms in MFMA fragment order (128 VGPRs per
// lane at H = 512); the cell state of a (row, unit) pair lives in one thread's register.  Per step a workgroup reads
// the 16 x H previous hidden rows of its row block (two bf16 planes, 32 KB), issues hi.hi + lo.hi + hi.lo on
// v_mfma_f32_16x16x32_bf16 (the products of an fp32 GEMM to ~2^-17), adds the input projection (computed for every step
// at once by the caller's GEMM), applies the cell and publishes its 16 x 16 block of the new hidden state.  The H/16
// workgroups of a (direction, row block) exchange hidden states through global memory with agent-scope (L2-bypassing)
// stores and loads -- the workgroups sit on different XCDs, whose L2s are not coherent with each other -- and WITHOUT
// flags or cache-maintenance fences: a value is one word that is either EMPTY or final (see pack_terms); different
// (direction, row block) groups never wait for each other.  All workgroups must be resident at
// once: the host entry refuses grids larger than the CU count, and a waiting wave gives up after ~2 s and raises the
// error word instead of hanging the device.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pika_las.h"
#include "pika_rnnt.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RB = 16;    // batch rows per workgroup (one MFMA row tile)
constexpr int UB = 16;    // hidden units per workgroup (x 4 gates = 4 MFMA column tiles)

__device__ inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float tanh_fast(float x) { return 1.0f - 2.0f * rcp(__expf(2.0f * x) + 1.0f); }
__device__ inline float sigmoid_fast(float x) { return rcp(1.0f + __expf(-x)); }

// packed[d][g][wave][gate][kt][plane][lane] x 8 bf16: the B fragment (column = unit g*16 + (lane & 15), reduction index
// (wave*KTW + kt)*32 + (lane >> 4)*8 + e) of gate `gate`, plane 0 = bf16(w), plane 1 = bf16(w - plane 0)
__global__ __launch_bounds__(256) void blstm_pack_kernel(const float *__restrict__ w, int D, int H, bf16x8 *__restrict__ packed) {
    const int KTW = H >> 7;
    const long long frag = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // (d, g, wave, gate, kt)
    const long long nfrag = (long long)D * (H / UB) * 4 * 4 * KTW;
    if (frag >= nfrag) return;
    const int lane = threadIdx.x & 63;
    long long f = frag;
    const int kt = (int)(f % KTW); f /= KTW;
    const int gate = (int)(f % 4); f /= 4;
    const int wave = (int)(f % 4); f /= 4;
    const int g = (int)(f % (H / UB));
    const int d = (int)(f / (H / UB));
    const float *src = w + ((long long)d * 4 * H + (long long)gate * H + g * UB + (lane & 15)) * H
                         + (wave * KTW + kt) * 32 + (lane >> 4) * 8;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = src[e];
        hi[e] = (__bf16)v;
        lo[e] = (__bf16)(v - (float)hi[e]);
    }
    packed[(frag * 2 + 0) * 64 + lane] = hi;
    packed[(frag * 2 + 1) * 64 + lane] = lo;
}

struct BL {
    const float *gx;        // (S, B, D*4H)
    const bf16x8 *w;
    const int *lens;        // (B,)
    float *out;             // (S, B, D*H)
    float *h_n, *c_n;       // (D, B, H)
    unsigned *xbuf;         // [D*nq][S][H / 8][RB][8] words (bf16 hi << 16 | bf16 lo), all EMPTY before the launch
    int *err;
    int S, B, D, H, nq, ng;
};

// A hidden value travels as ONE 32-bit word holding both bf16 terms, and a word that has not been written yet holds
// EMPTY (a NaN pattern no finite h produces): the consumer polls the data itself, there is no flag and no fence between
// "data written" and "flag written" -- one store-to-load round trip per step instead of two.  Every step has its own slot
// (S slots per group, set to EMPTY by a memset in front of the launch), so nothing is ever re-armed inside the kernel.
constexpr unsigned EMPTY = 0xffffffffu;

__device__ inline unsigned pack_terms(float h) {
    const __bf16 hi = (__bf16)h;
    const __bf16 lo = (__bf16)(h - (float)hi);
    const unsigned w = ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | __builtin_bit_cast(unsigned short, lo);
    return w == EMPTY ? 0xfffffffeu : w;        // a NaN state stays a NaN, and the peers do not wait for it forever
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16-byte loads at agent scope (`sc1`: what __hip_atomic_load compiles to, which exists for 4 and 8 bytes only).  The compiler
// does not count these loads: wait_loads names the registers they fill (csrc/lstm_train.hip: the same pair).
__device__ inline void load16_agent(u32x4 &d, const unsigned *p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ inline void wait_loads(u32x4 (&v)[N]) {
    static_assert(N == 2 || N == 4 || N == 6 || N == 8, "registers named one by one");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
    if constexpr (N == 6)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])::"memory");
    if constexpr (N == 8)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
}

template <int KTW>
__global__ __launch_bounds__(256) void blstm_layer_kernel(BL p) {
    __shared__ float part[2][4][4][RB * UB];        // [step parity][wave (reduction quarter)][gate][row][unit]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive blocks of a group land on the same XCD (block b runs on XCD b % 8) when there are 8 groups
    const int ngroups = p.D * p.nq;
    const int group = blockIdx.x % ngroups, g = blockIdx.x / ngroups;
    const int d = group / p.nq, q = group - d * p.nq;
    const int H = p.H, B = p.B, S = p.S;
    unsigned *xb = p.xbuf + (long long)group * S * RB * H;

    // recurrent weights of this wave's reduction quarter: 4 gates x KTW k-tiles x 2 planes, resident for the sequence
    bf16x8 wr[4][KTW][2];
    {
        const bf16x8 *wp = p.w + ((((long long)d * p.ng + g) * 4 + wave) * 4 * KTW * 2) * 64 + lane;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wr[gate][kt][pl] = wp[((gate * KTW + kt) * 2 + pl) * 64];
    }
    // cell ownership: thread -> (row r of the block, unit u of the block)
    const int r = tid >> 4, u = tid & 15;
    const int row = q * RB + r, unit = g * UB + u;
    const int len = row < B ? min(max(p.lens[row], 0), S) : 0;
    int nsteps = 0;
    for (int i = 0; i < RB; ++i) {
        const int rr = q * RB + i;
        if (rr < B) nsteps = max(nsteps, min(max(p.lens[rr], 0), S));
    }
    float c = 0.f, h = 0.f;
    const long long ldg = (long long)p.D * 4 * H, ldo = (long long)p.D * H;

    for (int s = 0; s < nsteps; ++s) {
        const bool valid = s < len;
        const int t = d == 0 ? s : len - 1 - s;
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {        // the input projection of this (t, row): independent of the recurrence, requested first
            const float *gp = p.gx + ((long long)t * B + row) * ldg + (long long)d * 4 * H + unit;
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) z[gate] = gp[(long long)gate * H];
        }
        f32x4 acc[4];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) acc[gate] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            // the 16 x (KTW*32) words of step s-1 this wave reduces over, in FRAGMENT order ([group of 8 units][row][8]: a lane
            // finds the 8 words of a k-tile in 32 contiguous bytes, the wave reads 2 KB in one piece per k-tile -- round 6:
            // as [row][H] words read 8 bytes at a time a wave instruction touched 64 sectors for 8 bytes each);
            // agent-scope loads (the writers sit on other XCDs), repeated until no word is EMPTY
            const unsigned *xs = xb + (long long)(s - 1) * RB * H + (((wave * KTW) * 4 + (lane >> 4)) * RB + (lane & 15)) * 8;
            u32x4 wq[KTW * 2];
            u32x2 wd[KTW][4];
            unsigned long long t0 = 0;
            for (unsigned spin = 0;; ++spin) {
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) {
                    load16_agent(wq[2 * kt], xs + kt * 4 * RB * 8);
                    load16_agent(wq[2 * kt + 1], xs + kt * 4 * RB * 8 + 4);
                }
                wait_loads(wq);
                bool empty = false;
#pragma unroll
                for (int i = 0; i < KTW * 2; ++i)
                    empty |= wq[i][0] == EMPTY || wq[i][1] == EMPTY || wq[i][2] == EMPTY || wq[i][3] == EMPTY;
                if (!__any(empty)) break;
                if ((spin & 63) == 63) {
                    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
                    const unsigned long long now = wall_clock64();          // 100 MHz
                    if (!t0) t0 = now;
                    else if (now - t0 > 200000000ull) {                     // 2 s: a peer is not running -- give up
                        __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) wd[kt][j] = u32x2{wq[2 * kt + (j >> 1)][2 * (j & 1)], wq[2 * kt + (j >> 1)][2 * (j & 1) + 1]};
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                u32x4 hb, lb;       // bf16 pairs: element 2i in the low half
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w0 = wd[kt][j].x, w1 = wd[kt][j].y;
                    hb[j] = (w0 >> 16) | (w1 & 0xffff0000u);
                    lb[j] = (w0 & 0xffffu) | (w1 << 16);
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, hb), al = __builtin_bit_cast(bf16x8, lb);
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) {
                    acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wr[gate][kt][1], acc[gate], 0, 0, 0);
                    acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wr[gate][kt][0], acc[gate], 0, 0, 0);
                    acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wr[gate][kt][0], acc[gate], 0, 0, 0);
                }
            }
        }
        // lane holds rows (lane >> 4)*4 + e of column lane & 15
        float (*pt)[4][RB * UB] = part[s & 1];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int e = 0; e < 4; ++e) pt[wave][gate][((lane >> 4) * 4 + e) * UB + (lane & 15)] = acc[gate][e];
        __syncthreads();        // the only barrier of a step: `part` alternates, step s+1 writes the other half
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int w = 0; w < 4; ++w) z[gate] += pt[w][gate][tid];
        if (valid) {
            c = sigmoid_fast(z[1]) * c + sigmoid_fast(z[0]) * tanh_fast(z[2]);
            h = sigmoid_fast(z[3]) * tanh_fast(c);
        }
        if (s + 1 < nsteps)     // rows past their end publish their frozen state: the peers only test for EMPTY
            __hip_atomic_store(xb + (long long)s * RB * H + ((unit >> 3) * RB + r) * 8 + (unit & 7), pack_terms(h), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (valid) {
            p.out[((long long)t * B + row) * ldo + (long long)d * H + unit] = h;
            if (s == len - 1) {
                p.h_n[((long long)d * B + row) * H + unit] = h;
                p.c_n[((long long)d * B + row) * H + unit] = c;
            }
        }
    }
    // padded positions of the rows: zeros, as pad_packed_sequence returns them
    if (row < B) {
        for (int t = len; t < S; ++t) p.out[((long long)t * B + row) * ldo + (long long)d * H + unit] = 0.f;
        if (len == 0) {
            p.h_n[((long long)d * B + row) * H + unit] = 0.f;
            p.c_n[((long long)d * B + row) * H + unit] = 0.f;
        }
    }
}

inline int geometry(int B, int D, int H, int *nq, int *ng) {
    if (B <= 0 || (D != 1 && D != 2) || H <= 0) return PIKA_EINVAL;
    if ((H & 127) || H > 512) return PIKA_ETOOBIG;           // 128 registers of weights per lane at H = 512
    *nq = (B + RB - 1) / RB;
    *ng = H / UB;
    return PIKA_OK;
}

}  // namespace

extern "C" {

long long pika_blstm_packed_bytes(int D, int H) {
    int nq, ng;
    if (geometry(1, D, H, &nq, &ng) != PIKA_OK) return -1;
    return (long long)D * 4 * H * H * 2 * 2;
}

long long pika_blstm_work_bytes(int S, int B, int D, int H) {
    int nq, ng;
    if (geometry(B, D, H, &nq, &ng) != PIKA_OK || S <= 0) return -1;
    return 256 + (long long)D * nq * S * RB * H * 4;        // [error word | pad][one slot of 16 x H words per step and group]
}

int pika_blstm_pack(const float *w_hh, int D, int H, void *packed, void *stream) {
    int nq, ng;
    const int rc = geometry(1, D, H, &nq, &ng);
    if (rc != PIKA_OK) return rc;
    if (!w_hh || !packed || (reinterpret_cast<uintptr_t>(packed) & 15)) return PIKA_EINVAL;
    const long long nfrag = (long long)D * ng * 4 * 4 * (H >> 7);
    hipLaunchKernelGGL(blstm_pack_kernel, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w_hh, D, H, static_cast<bf16x8 *>(packed));
    return (int)hipGetLastError();
}

int pika_blstm_layer(const float *gx, const void *w_packed, const int *lens, float *out, float *h_n, float *c_n,
                     void *work, long long work_bytes, int S, int B, int D, int H, void *stream) {
    int nq, ng;
    const int rc = geometry(B, D, H, &nq, &ng);
    if (rc != PIKA_OK) return rc;
    if (!gx || !w_packed || !lens || !out || !h_n || !c_n || !work || S <= 0) return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(work)) & 15) return PIKA_EINVAL;
    if (work_bytes < pika_blstm_work_bytes(S, B, D, H)) return PIKA_EINVAL;
    // every workgroup waits for others inside the kernel: the whole grid has to be resident
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return PIKA_EINVAL;
    const int grid = D * nq * ng;
    if (grid > cus) return PIKA_ETOOBIG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    BL p;
    p.gx = gx; p.w = static_cast<const bf16x8 *>(w_packed); p.lens = lens; p.out = out; p.h_n = h_n; p.c_n = c_n;
    p.err = static_cast<int *>(work);
    p.xbuf = reinterpret_cast<unsigned *>(static_cast<char *>(work) + 256);
    p.S = S; p.B = B; p.D = D; p.H = H; p.nq = nq; p.ng = ng;
    hipError_t e = hipMemsetAsync(work, 0, 256, st);
    if (e == hipSuccess) e = hipMemsetAsync(p.xbuf, 0xff, (size_t)D * nq * S * RB * H * 4, st);
    if (e != hipSuccess) return (int)e;
    switch (H >> 7) {
    case 1: hipLaunchKernelGGL(blstm_layer_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL(blstm_layer_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL(blstm_layer_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(blstm_layer_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
    }
    return (int)hipGetLastError();
}

int pika_blstm_status(const void *work, int *host_out, void *stream) {
    if (!work || !host_out) return PIKA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(host_out, work, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return (int)e;
}

}  // extern "C"
