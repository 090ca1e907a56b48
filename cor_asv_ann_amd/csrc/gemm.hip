// fp32 MFMA GEMM for gfx950 with an optional fused LSTM-cell epilogue.
//
//   PLAIN:  C[M][N]  = A[M][K] . Bt[N][K]^T (+ bias)
//   LSTM:   z = [x | ctx | h][M][K] . Wt[4U][K]^T + b ; (h', c') = cell(z, c)     (U = units)
//
// The A operand is the concatenation of up to three row-gathered K-segments (layer input, attention
// context, recurrent state), so "x.K + h.R" of a Keras LSTMCell (seq2seq.py:272,337,344) is ONE
// contraction and the gate pre-activations never leave the accumulators.  Weight rows are stored
// gate-interleaved in blocks of 32 units ([unit/32][gate i,f,c,o][unit%32]) so that a 128-column block
// tile holds all four gates of 32 units and a wave's four 32x32 accumulator tiles hold, register for
// register, the i/f/c/o pre-activations of the same (row, unit).
//
// Tiling: 128x128 block tile, BK = 32, 256 threads = 4 waves, each wave 32 rows x 128 columns
// (4 x v_mfma_f32_32x32x2_f32 accumulators, exact fp32 = a k-ordered fmaf chain).  Operands are staged
// K-contiguous in LDS with a +4-float row pad (144-B rows: ds_read_b128 is conflict-free for the 16-lane
// groups) and double-buffered; each lane reads four consecutive k per ds_read_b128 and feeds them to
// four MFMAs (the k order inside a tile is permuted identically for A and B, which a sum allows).
#include "common.h"

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, LDW = BK + 4;
constexpr int TILE_FLOATS = 128 * LDW;               // one operand tile
constexpr int GEMM_LDS_BYTES = 2 * 2 * TILE_FLOATS * 4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int step = g.step_ptr ? *g.step_ptr : g.step_imm;
    const int nbn = (g.N + BN - 1) / BN;
    const int bn = blockIdx.x % nbn, bm = blockIdx.x / nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    if (m0 >= g.M) return;

    const int r0 = tid >> 3, kc = tid & 7;

    // per-segment row pointers of the four rows this thread stages (statically indexed: no scratch)
    const float* ap0[4]; const float* ap1[4]; const float* ap2[4];
    int tiles0 = 0, tiles1 = 0, tiles2 = 0;
#define CASV_SETUP_SEG(S, AP, TILES)                                                             \
    if (g.nseg > S && !(g.a[S].skip_first && step == 0)) {                                       \
        const Seg& sg = g.a[S];                                                                  \
        const float* base = sg.base + (long long)(step * sg.step_mul + sg.step_add) * sg.slot_stride; \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
            int m = m0 + r0 + 32 * i; m = m < g.M ? m : g.M - 1;                                 \
            const int rid = sg.rows ? sg.rows[m] : m;                                            \
            AP[i] = base + (long long)rid * sg.ld + 4 * kc;                                      \
        }                                                                                        \
        TILES = sg.width / BK;                                                                   \
    } else {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) AP[i] = nullptr;                           \
    }
    CASV_SETUP_SEG(0, ap0, tiles0)
    CASV_SETUP_SEG(1, ap1, tiles1)
    CASV_SETUP_SEG(2, ap2, tiles2)
#undef CASV_SETUP_SEG
    const int c0 = tiles0, c1 = c0 + tiles1, ntiles = c1 + tiles2;

    const float* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = n0 + r0 + 32 * i; n = n < g.N ? n : g.N - 1;
        bp[i] = g.Bt + (long long)n * g.Ktot + 4 * kc;
    }

    f32x4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
        int kb;
        if (kt < c0) {
            const int ko = kt * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap0[i] + ko);
            kb = g.a[0].koff + ko;
        } else if (kt < c1) {
            const int ko = (kt - c0) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap1[i] + ko);
            kb = g.a[1].koff + ko;
        } else {
            const int ko = (kt - c1) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap2[i] + ko);
            kb = g.a[2].koff + ko;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + kb);
    };
    auto store_tile = [&](int buf) {
        float* sa = smem + buf * 2 * TILE_FLOATS + r0 * LDW + 4 * kc;
        float* sb = sa + TILE_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(sa + 32 * i * LDW) = ra[i];
            *reinterpret_cast<f32x4*>(sb + 32 * i * LDW) = rb[i];
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

    if (ntiles > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + 1 < ntiles;
        if (more) load_tile(kt + 1);
        const float* As = smem + buf * 2 * TILE_FLOATS + (wave * 32 + l31) * LDW + 4 * lh;
        const float* Bs = smem + buf * 2 * TILE_FLOATS + TILE_FLOATS + l31 * LDW + 4 * lh;
        f32x4 af[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f32x4*>(As + 8 * j);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 bf = *reinterpret_cast<const f32x4*>(Bs + c * 32 * LDW + 8 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][i], bf[i], acc[c], 0, 0, 0);
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue ----
    if (EPI == EPI_PLAIN) {
        float* cbase = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + c * 32 + l31;
            if (n < g.N) {
                const float b = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < g.M) cbase[(long long)m * g.out.ld + n] = acc[c][r] + b;
                }
            }
        }
    } else {
        const int u = bn * 32 + l31;     // hidden unit of this lane
        const float bi = g.bias[n0 + l31], bf_ = g.bias[n0 + 32 + l31];
        const float bg = g.bias[n0 + 64 + l31], bo = g.bias[n0 + 96 + l31];
        const bool czero = g.c_in.skip_first && step == 0;
        const float* cin = g.c_in.base + (long long)(step * g.c_in.step_mul + g.c_in.step_add) * g.c_in.slot_stride;
        float* cout = g.c_out.base + (long long)(step * g.c_out.step_mul + g.c_out.step_add) * g.c_out.slot_stride;
        float* hout = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < g.M) {
                float cprev = 0.0f;
                if (!czero) {
                    const int rid = g.c_in.rows ? g.c_in.rows[m] : m;
                    cprev = cin[(long long)rid * g.c_in.ld + u];
                }
                const float ig = sigmoidf_(acc[0][r] + bi);
                const float fg = sigmoidf_(acc[1][r] + bf_);
                const float gg = tanhf(acc[2][r] + bg);
                const float og = sigmoidf_(acc[3][r] + bo);
                const float c2 = fg * cprev + ig * gg;
                const float h2 = og * tanhf(c2);
                cout[(long long)m * g.c_out.ld + u] = c2;
                hout[(long long)m * g.out.ld + u] = h2;
            }
        }
    }
}

void launch_gemm(int epi, const GemmArgs& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<EPI_PLAIN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<EPI_LSTM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        attr_set = true;
    }
    const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
    const dim3 grid(nbm * nbn), block(256);
    if (epi == EPI_LSTM)
        hipLaunchKernelGGL(gemm_kernel<EPI_LSTM>, grid, block, GEMM_LDS_BYTES, stream, g);
    else
        hipLaunchKernelGGL(gemm_kernel<EPI_PLAIN>, grid, block, GEMM_LDS_BYTES, stream, g);
}

}  // namespace casv
