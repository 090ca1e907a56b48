// The forward recurrences of the train step (keras LSTM over time, seq2seq.py:268-272,293-298,363-367) as ONE launch per
// pair of layers instead of one launch per time step.
//
// train.hip walks encoder layer n and decoder layer n-1 in lockstep: per step one gemm_skinny launch of 2 x 256 workgroups,
// each contracting a 32 x 128 tile of h(t-1).Wr^T over K = W and finishing the cell.  A launch lasts ~30 us of which the
// matrix pipes need ~15; the rest is launch gap, the first round trip to the L2 and the tail.  Here the same workgroups stay
// resident for the whole sequence: workgroup (layer, row block, unit group) owns 32 rows x 32 units, keeps their cell state in
// registers, and hands h(t) to the 16 workgroups of its row block through memory (handoff.h: sc1 stores + drain + one counter
// per row block; sc1 loads behind the poll).  The weights of a unit group (128 rows of Wr) are read by 2 x 16 workgroups; the
// mapping puts a unit group's workgroups on ONE XCD, whose L2 then holds 2 groups x 2 layers x 256 KB.
//
// What a workgroup does per step, in the order that hides the most: the first three weight stages and the step's x.Wx + b
// values are requested BEFORE it polls (they do not depend on h); behind the poll all W/32 stages of its 32 rows of h(t-1)
// leave at once (16 B per thread and stage: one memory round trip per step, not one per stage -- these lines were written by
// other XCDs a moment ago and miss every cache); then gemm_skinny.hip's K loop, stage by stage through the same LDS image with
// the same MFMA sequence, and its cell epilogue.  Results are therefore those of the per-step launches bit for bit (tested).
#include "common.h"
#include "handoff.h"
#include "train_kernels.h"
#include <map>
#include <mutex>

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int RBM = 32, RBN = 128;
constexpr int RK2 = 32, RLD = RK2 + 4;
constexpr int RSTAGE = (RBM + RBN) * RLD;
}

template <int NT>        // W / 32: K stages per step
__global__ __launch_bounds__(256, 2) void train_recurrence_kernel(const RecArgs ra) {
    __shared__ __attribute__((aligned(16))) float s_stage[2 * RSTAGE];
    __shared__ int s_ok;
    float (*s_gate)[16][64] = reinterpret_cast<float (*)[16][64]>(s_stage);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int W = NT * 32;
    const int B = ra.B;
    const int nrb = (B + RBM - 1) / RBM;
    // consecutive workgroups go to consecutive XCDs: unit group fastest, so that XCD x sees unit groups x, x + 8, ...
    const int ug = blockIdx.x % NT, rest = blockIdx.x / NT;
    const int rb = rest % nrb, jb = rest / nrb;
    const RecJob& job = ra.job[jb];
    const int m0 = rb * RBM, n0 = ug * RBN;
    unsigned* const counter = ra.counters + (long long)(jb * nrb + rb) * 32;
    unsigned* const abort_w = ra.counters + (long long)2 * nrb * 32;
    const int len = job.len;

    const int srow = tid >> 3, sk = 4 * (tid & 7);
    int mrow = m0 + srow; mrow = mrow < B ? mrow : B - 1;
    const float* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bp[i] = job.Wr + (long long)(n0 + srow + 32 * i) * W + sk;

    struct BStage { f32x4 b[4]; };
    // stage KT of the weights / of the rows of h: the stage's offset rides in the instruction (no address registers per stage)
#define CASV_LOAD_B(G, KT)                                                                                              \
    {                                                                                                                   \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[0]) : "v"(bp[0]), "n"((KT) * RK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[1]) : "v"(bp[1]), "n"((KT) * RK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[2]) : "v"(bp[2]), "n"((KT) * RK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[3]) : "v"(bp[3]), "n"((KT) * RK2 * 4));      \
    }
#define CASV_LOAD_A(J)                                                                                                  \
    if constexpr ((J) < NT) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(areg[(J) < NT ? (J) : 0]) : "v"(arow), "n"((J) * RK2 * 4));
    auto store_b = [&](const BStage& gs, int buf) {
        float* sb = s_stage + buf * RSTAGE + (RBM + srow) * RLD + sk;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sb + 32 * i * RLD) = gs.b[i];
    };
    auto store_a = [&](const f32x4& a, int buf) {
        *reinterpret_cast<f32x4*>(s_stage + buf * RSTAGE + srow * RLD + sk) = a;
    };
    const int a_off = l31 * RLD + 4 * lh, b_off = (RBM + wave * 32 + l31) * RLD + 4 * lh;
    f32x16 acc;
    auto compute = [&](int buf) {
        const float* base = s_stage + buf * RSTAGE;
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[q] = *reinterpret_cast<const f32x4*>(base + a_off + 8 * q);
            fb[q] = *reinterpret_cast<const f32x4*>(base + b_off + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i], fb[q][i], acc, 0, 0, 0);
    };

    // cell state of this lane's four (row, unit) elements: rows m0 + q + 8 wave + 4 lh, unit ug * 32 + l31
    const int u = ug * 32 + l31;
    const float omask_u = job.omask ? job.omask[u] : 1.0f;
    float cst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int m = m0 + q + 8 * wave + 4 * lh; m = m < B ? m : B - 1;
        cst[q] = job.c0 ? job.c0[(long long)m * W + u] : 0.0f;
    }

    for (int k = 0; k < len; ++k) {
        if (ra.fault && blockIdx.x == 0 && k == 1) return;      // (test: a workgroup that never hands on -- its peers must give up)
        const int t = job.reverse ? len - 1 - k : k, tp = job.reverse ? t + 1 : t - 1;
        const float* arow = k == 0 ? (job.h0 ? job.h0 + (long long)mrow * W + sk : nullptr)
                                   : job.hs + ((long long)tp * B + mrow) * job.hs_ld + sk;
        // x.Wx + b of this step's elements
        float zpre[4][4];
        {
            const float* zin0 = job.Z + (long long)t * B * (4 * W);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + q + 8 * wave + 4 * lh;
                const float* zr = zin0 + (long long)(m < B ? m : B - 1) * (4 * W) + n0 + l31;
                zpre[q][0] = zr[0]; zpre[q][1] = zr[32]; zpre[q][2] = zr[64]; zpre[q][3] = zr[96];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        if (arow) {         // (uniform: a zero initial state contributes nothing -- the per-step path skips the segment too)
            BStage g0, g1;
            CASV_LOAD_B(g0, 0)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g0.b[2]), "+v"(g0.b[3]));
            store_b(g0, 0);
            if constexpr (NT > 1) CASV_LOAD_B(g1, 1)
            if constexpr (NT > 2) CASV_LOAD_B(g0, 2)
            if (k > 0) {
                if (!wait_deps(Dep{counter, (unsigned)(k * NT)}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
            }
            f32x4 areg[NT];
            CASV_LOAD_A(0) CASV_LOAD_A(1) CASV_LOAD_A(2) CASV_LOAD_A(3) CASV_LOAD_A(4) CASV_LOAD_A(5) CASV_LOAD_A(6) CASV_LOAD_A(7)
            CASV_LOAD_A(8) CASV_LOAD_A(9) CASV_LOAD_A(10) CASV_LOAD_A(11) CASV_LOAD_A(12) CASV_LOAD_A(13) CASV_LOAD_A(14) CASV_LOAD_A(15)
            // everything requested so far has arrived
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg[j]));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g0.b[2]), "+v"(g0.b[3]),
                                                "+v"(g1.b[0]), "+v"(g1.b[1]), "+v"(g1.b[2]), "+v"(g1.b[3]));
            store_a(areg[0], 0);
            __syncthreads();
            // stage J + 1 goes from registers into the LDS buffer whose readers passed the last barrier, stage J + 3 leaves for the
            // same register set, stage J is contracted; the wait in front of the store is counted (stage J + 2 stays in flight)
#define CASV_REC_STAGE(G, J)                                                                                            \
            if constexpr ((J) < NT) {                                                                                   \
                if constexpr ((J) + 1 < NT) {                                                                           \
                    if constexpr ((J) >= 2 && (J) + 2 < NT)                                                             \
                        asm volatile("s_waitcnt vmcnt(4)" : "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3]));    \
                    else if constexpr ((J) >= 2)                                                                        \
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3]));    \
                    store_a(areg[(J) + 1 < NT ? (J) + 1 : 0], ((J) + 1) & 1);                                           \
                    store_b(G, ((J) + 1) & 1);                                                                          \
                }                                                                                                       \
                if constexpr ((J) + 3 < NT) CASV_LOAD_B(G, (J) + 3)                                                     \
                compute((J) & 1);                                                                                       \
                __syncthreads();                                                                                        \
            }
            CASV_REC_STAGE(g1, 0) CASV_REC_STAGE(g0, 1) CASV_REC_STAGE(g1, 2) CASV_REC_STAGE(g0, 3)
            CASV_REC_STAGE(g1, 4) CASV_REC_STAGE(g0, 5) CASV_REC_STAGE(g1, 6) CASV_REC_STAGE(g0, 7)
            CASV_REC_STAGE(g1, 8) CASV_REC_STAGE(g0, 9) CASV_REC_STAGE(g1, 10) CASV_REC_STAGE(g0, 11)
            CASV_REC_STAGE(g1, 12) CASV_REC_STAGE(g0, 13) CASV_REC_STAGE(g1, 14) CASV_REC_STAGE(g0, 15)
#undef CASV_REC_STAGE
        } else if (k > 0) {
            return;          // (never: only the first step can lack its input)
        }

        // ---- cell epilogue (gemm_skinny.hip) ----
#pragma unroll
        for (int r = 0; r < 16; ++r) s_gate[wave][r][lane] = acc[r];
        __syncthreads();
        float z[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) z[q][c] = s_gate[c][4 * wave + q][lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) { z[q][0] += zpre[q][0]; z[q][1] += zpre[q][1]; z[q][2] += zpre[q][2]; z[q][3] += zpre[q][3]; }
        float* cout = job.Cs + (long long)t * B * W;
        float* hout = job.hs + (long long)t * B * job.hs_ld;
        float* gout = job.Gt + (long long)t * B * (4 * W);
        float* mout = job.om ? job.om + (long long)t * B * job.om_ld : nullptr;        // the outputs once more, masked: the next layer's input
        // (the layer's dL/dh accumulator of the backward pass, cleared here 16 bytes per thread and step instead of by a fill of the
        // whole buffer in front of the backward pass: these stores cost nothing beside the epilogue's own)
        if (job.zero && m0 + srow < B)
            *reinterpret_cast<f32x4*>(job.zero + ((long long)t * B + m0 + srow) * W + ug * 32 + sk) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + q + 8 * wave + 4 * lh;
            if (m < B) {
                const LstmCellOut cell = lstm_cell(z[q][0] + 0.f, z[q][1] + 0.f, z[q][2] + 0.f, z[q][3] + 0.f, cst[q]);
                cst[q] = cell.c;
                cout[(long long)m * W + u] = cell.c;
                __hip_atomic_store(hout + (long long)m * job.hs_ld + u, cell.h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (mout) mout[(long long)m * job.om_ld + u] = cell.h * omask_u;
                float* gr = gout + (long long)m * (4 * W) + n0 + l31;
                gr[0] = cell.i; gr[32] = cell.f; gr[64] = cell.g; gr[96] = cell.o;
            }
        }
        publish(counter);        // drain + barrier (also: the gate exchange is read before the next step's stage lands on it)
    }
}

template <class K>
static int rec_blocks_per_cu(K kernel) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(mu);
    const void* f = reinterpret_cast<const void*>(kernel);
    auto it = cache.find({dev, f});
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, 0) != hipSuccess) n = 0;
    n = n > 2 ? 2 : (n < 0 ? 0 : n);
    cache[{dev, f}] = n;
    return n;
}

size_t train_recurrence_counter_bytes(int B) { return ((size_t)2 * ((B + RBM - 1) / RBM) * 32 + 32) * sizeof(unsigned); }

// Workgroups of the launch if this shape has a persistent form on a device of `ncu` CUs whose every workgroup is resident at
// once, else 0 (the caller runs the per-step launches).
template <int NT> static int rec_grid(const RecArgs& ra, int ncu) {
    const int grid = ra.njobs * ((ra.B + RBM - 1) / RBM) * NT;
    return grid <= rec_blocks_per_cu(train_recurrence_kernel<NT>) * ncu ? grid : 0;
}
#define CASV_REC_WIDTHS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
int train_recurrence_grid(const RecArgs& ra, int ncu) {
    if (ra.W % 32 || ra.njobs < 1 || ra.njobs > 2 || ra.B < 1) return 0;
    switch (ra.W / 32) {
#define CASV_REC_CASE(NT_) case NT_: return rec_grid<NT_>(ra, ncu);
        CASV_REC_WIDTHS(CASV_REC_CASE)
#undef CASV_REC_CASE
        default: return 0;          // wider layers: the rows of h no longer fit a thread's registers
    }
}

void launch_train_recurrence(const RecArgs& ra, int grid, hipStream_t stream) {
    switch (ra.W / 32) {
#define CASV_REC_CASE(NT_) case NT_: hipLaunchKernelGGL((train_recurrence_kernel<NT_>), dim3(grid), dim3(256), 0, stream, ra); break;
        CASV_REC_WIDTHS(CASV_REC_CASE)
#undef CASV_REC_CASE
        default: break;
    }
}
#undef CASV_REC_WIDTHS

}  // namespace casv
