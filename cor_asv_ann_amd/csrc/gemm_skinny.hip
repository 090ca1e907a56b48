// Skinny-M variant of the fused GEMM (same GemmArgs, same results bit for bit as gemm.hip with KS = 1).
//
// A 128x128 tile grid gives ceil(M/128) * N/128 workgroups: at M = 512 (train step) that is 64 of them, at M = 256
// (greedy decode of a default batch) 32 -- on 256 CUs, each CU's matrix pipe then works through a whole 128x128xK tile
// while most of the chip idles.  Here a workgroup owns a 32 x 128 tile (4x as many workgroups) and its four waves
// own one 32x32 accumulator tile each -- wave w = gate w of the LSTM epilogue -- over the FULL K range, in the same
// k order and with the same MFMA sequence per element as the big kernel (so a row's value does not depend on which
// variant ran).  There is no operand reuse between the waves of a workgroup beyond the 32-row A panel (served by the
// L1), so fragments go straight from global memory into MFMA operand registers, four K-tiles deep; the LDS is used
// once, to bring the four gates of a (row, unit) into one lane for the cell epilogue.
#include "common.h"

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SBM = 32, SBN = 128, SBK = 16, SDEPTH = 4;      // prefetch depth 4 = 6 = 8 K-tiles (measured; in round 2 again: 8 gains 2 % on the train step, nothing on decode)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(const GemmBatch batch) {
    __shared__ float s_gate[4][16][64];
    const GemmArgs& g = batch.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int step = __builtin_amdgcn_readfirstlane(g.step_ptr ? *g.step_ptr : g.step_imm);
    const int nbn = (g.N + SBN - 1) / SBN;
    const int bn = blockIdx.x % nbn, bm = blockIdx.x / nbn;      // consecutive workgroups (= different XCDs) take different B panels
    const int m0 = bm * SBM, n0 = bn * SBN;
    if (m0 >= g.M) return;
    if (g.nact) {           // every wave looks at the same counts: a uniform exit ahead of the first barrier
        const int mlast = (m0 + SBM < g.M ? m0 + SBM : g.M) - 1;
        const int l0 = m0 / g.nact_group, l1 = mlast / g.nact_group;
        int alive = 0;
        for (int l = l0 + lane; l <= l1; l += 64) alive |= g.nact[l] > (l == l0 ? m0 - l0 * g.nact_group : 0);
        if (!__any(alive)) return;
    }

    // A: this lane's row of every segment (fragment layout: lane (l31, lh) holds k = 4*lh + 8*j + i of row l31)
    const float* ap0; const float* ap1; const float* ap2;
    int tiles0 = 0, tiles1 = 0, tiles2 = 0;
    int mrow = m0 + l31; mrow = mrow < g.M ? mrow : g.M - 1;
#define CASV_SETUP_SEG(S, AP, TILES)                                                             \
    if (g.nseg > S && !(g.a[S].skip_first && step == 0 && !g.a[S].first_base)) {                 \
        const Seg& sg = g.a[S];                                                                  \
        const bool first = sg.first_base && step == 0;                                           \
        const float* base = first ? sg.first_base                                                \
            : sg.base + (long long)(step * sg.step_mul + sg.step_add) * sg.slot_stride;          \
        const int rid = (sg.rows && !first) ? sg.rows[mrow] : mrow;                              \
        AP = base + (long long)rid * sg.ld + 4 * lh;                                             \
        TILES = sg.width / SBK;                                                                  \
    } else {                                                                                     \
        AP = nullptr;                                                                            \
    }
    CASV_SETUP_SEG(0, ap0, tiles0)
    CASV_SETUP_SEG(1, ap1, tiles1)
    CASV_SETUP_SEG(2, ap2, tiles2)
#undef CASV_SETUP_SEG
    const int c0 = __builtin_amdgcn_readfirstlane(tiles0), c1 = __builtin_amdgcn_readfirstlane(tiles0 + tiles1);
    const int ntiles_all = __builtin_amdgcn_readfirstlane(tiles0 + tiles1 + tiles2);
    const int nsplit = gridDim.z;
    const int per = (ntiles_all + nsplit - 1) / nsplit;
    const int kt_begin = blockIdx.z * per;
    const int ntiles = ntiles_all - kt_begin < per ? (ntiles_all - kt_begin > 0 ? ntiles_all - kt_begin : 0) : per;
    const int koff0 = g.a[0].koff, koff1 = g.a[1].koff, koff2 = g.a[2].koff;
    const long long d1 = (long long)((const char*)ap1 - (const char*)ap0), d2 = (long long)((const char*)ap2 - (const char*)ap0);

    // B: this wave's 32 columns (LSTM: gate `wave` of the 32 units of the tile)
    int ncol = n0 + wave * 32 + l31; ncol = ncol < g.N ? ncol : g.N - 1;
    const float* bp = g.Bt + (long long)ncol * g.Ktot + 4 * lh;

    struct Frag { f32x4 a[2], b[2]; };
    auto load = [&](Frag& f, int kt_rel) {
        int kt = kt_rel < ntiles ? kt_rel : ntiles - 1;         // past the end: a valid, unused re-load
        kt += kt_begin;
        const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
        const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1);
        const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * SBK;
        const char* pa = (const char*)ap0 + (d1 & m1) + (d2 & m2) + (long long)ko * (SBK * 4);
        f.a[0] = *reinterpret_cast<const f32x4*>(pa); f.a[1] = *reinterpret_cast<const f32x4*>(pa + 32);
        f.b[0] = *reinterpret_cast<const f32x4*>(bp + kb); f.b[1] = *reinterpret_cast<const f32x4*>(bp + kb + 8);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j][i], f.b[j][i], acc, 0, 0, 0);
    };

    // LSTM: wave w finishes rows (r & 3, r >> 2 == w) of the tile; fetch their previous cell state under the K loop
    float cpv[4];
    if (EPI == EPI_LSTM && !g.epi_plain) {
        const bool cfirst = g.c_in.first_base && step == 0;
        const bool czero = g.c_in.skip_first && step == 0 && !cfirst;
        const float* cin = cfirst ? g.c_in.first_base
            : g.c_in.base + (long long)(step * g.c_in.step_mul + g.c_in.step_add) * g.c_in.slot_stride;
        const int u = bn * 32 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int m = m0 + q + 8 * wave + 4 * lh;
            m = m < g.M ? m : g.M - 1;
            cpv[q] = 0.0f;
            if (!czero) {
                const int rid = (g.c_in.rows && !cfirst) ? g.c_in.rows[m] : m;
                cpv[q] = cin[(long long)rid * g.c_in.ld + u];
            }
        }
    }

    if (ntiles > 0) {
        Frag f[SDEPTH];
#pragma unroll
        for (int q = 0; q < SDEPTH; ++q) load(f[q], q);
        int kt = 0;
        for (; kt + SDEPTH <= ntiles; kt += SDEPTH) {
#pragma unroll
            for (int q = 0; q < SDEPTH; ++q) {
                mma(f[q]);
                load(f[q], kt + SDEPTH + q);
            }
        }
        const int rest = ntiles - kt;
#pragma unroll
        for (int q = 0; q < SDEPTH - 1; ++q)
            if (rest > q) mma(f[q]);
    }

    // ---- epilogue ----
    if (EPI == EPI_PLAIN || g.epi_plain) {
        float* cbase = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        const int n = n0 + wave * 32 + l31;
        if (n < g.N) {
            const float b = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < g.M) {
                    float* dst = cbase + (long long)m * g.out.ld + n;
                    if (nsplit > 1) atomicAdd(dst, acc[r] + (blockIdx.z == 0 ? b : 0.0f));
                    else *dst = g.accumulate ? (*dst + acc[r] + b) : (acc[r] + b);
                }
            }
        }
    } else {
        // gate w of every (row, unit) of the tile -> LDS; then wave w takes accumulator rows r = 4w .. 4w+3 of all four gates
#pragma unroll
        for (int r = 0; r < 16; ++r) s_gate[wave][r][lane] = acc[r];
        __syncthreads();
        const int u = bn * 32 + l31;
        const float bi = g.bias ? g.bias[n0 + l31] : 0.f, bf_ = g.bias ? g.bias[n0 + 32 + l31] : 0.f;
        const float bg = g.bias ? g.bias[n0 + 64 + l31] : 0.f, bo = g.bias ? g.bias[n0 + 96 + l31] : 0.f;
        const float* zin = g.zinit.base
            ? g.zinit.base + (long long)(step * g.zinit.step_mul + g.zinit.step_add) * g.zinit.slot_stride : nullptr;
        float* cout = g.c_out.base + (long long)(step * g.c_out.step_mul + g.c_out.step_add) * g.c_out.slot_stride;
        float* hout = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        float* gout = g.gates_out.base
            ? g.gates_out.base + (long long)(step * g.gates_out.step_mul + g.gates_out.step_add) * g.gates_out.slot_stride
            : nullptr;
        float z[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) z[q][c] = s_gate[c][4 * wave + q][lane];
        if (zin) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + q + 8 * wave + 4 * lh;
                const float* zr = zin + (long long)(m < g.M ? m : g.M - 1) * g.zinit.ld + n0 + l31;
                z[q][0] += zr[0]; z[q][1] += zr[32]; z[q][2] += zr[64]; z[q][3] += zr[96];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + q + 8 * wave + 4 * lh;
            if (m < g.M) {
                const LstmCellOut cell = lstm_cell(z[q][0] + bi, z[q][1] + bf_, z[q][2] + bg, z[q][3] + bo, cpv[q]);
                cout[(long long)m * g.c_out.ld + u] = cell.c;
                hout[(long long)m * g.out.ld + u] = cell.h;
                if (gout) {
                    float* gr = gout + (long long)m * g.gates_out.ld + n0 + l31;
                    gr[0] = cell.i; gr[32] = cell.f; gr[64] = cell.g; gr[96] = cell.o;
                }
            }
        }
    }
}

void launch_gemm_skinny(int epi, const GemmBatch& b, int ksplit, hipStream_t stream) {
    int blocks = 0;
    for (int j = 0; j < b.count; ++j) {
        const int nb = ((b.g[j].M + SBM - 1) / SBM) * ((b.g[j].N + SBN - 1) / SBN);
        blocks = nb > blocks ? nb : blocks;
    }
    if (epi == EPI_LSTM) hipLaunchKernelGGL((gemm_skinny_kernel<EPI_LSTM>), dim3(blocks, b.count, ksplit), dim3(256), 0, stream, b);
    else hipLaunchKernelGGL((gemm_skinny_kernel<EPI_PLAIN>), dim3(blocks, b.count, ksplit), dim3(256), 0, stream, b);
}

}  // namespace casv
