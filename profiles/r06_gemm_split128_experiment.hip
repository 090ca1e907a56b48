// The fused LSTM-cell GEMM of gemm_split.hip (bf16x3-split operands) as 128 x 256 tiles: 4 waves per
// workgroup, each wave the SAME 64 rows x 128 columns (all four gates of 32 units) and the same product sequence per element as in the
// 256 x 256 kernel -- the same bits -- but TWO workgroups per CU (2 x 72 KB of tile buffers) instead of one.  Why: with one workgroup
// per CU and one round of workgroups, a launch's prologue, its epilogue (the LSTM cells of the tile) and the spread between the CUs'
// run times are exposed -- 3.4 + 7.4 + 9.8 us of 147 at K = 1024 (profiles/r06_split_stamps.txt).  Two workgroups that share a CU
// fall out of step by themselves (the matrix pipe serves the older wave first: profiles/r04_gemm_stamps.txt section 5), so one's
// cells and stores run under the other's products, and each has half the cells.
//
// Weights come from the pre-split LDS-layout image of gemm_split.hip (the SAME image: its tiles are 256 columns wide) by LDS-DMA;
// only A is staged through registers (split while stored).  The previous cell state is read straight from memory in the epilogue
// (no room for it in the LDS).  Compiler-scheduled loop, one barrier per K tile:
//   fragments of tile t <- LDS[t & 1];  B planes of tile t + 1 -> LDS[(t + 1) & 1] (LDS-DMA);  48 products;
//   wait (the transfers; the A rows of tile t + 1, requested one tile ago);  A of tile t + 1 split and stored;  A of tile t + 2 requested.
// Jobs: fused LSTM epilogue, static weights with an image, K segments in whole tiles -- what the beam search's layer launches are;
// everything else stays with gemm_split.hip.
#include "common.h"
#include <cstdlib>

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int S1_BM = 128, S1_BN = 256, S1_BK = 16;
constexpr int S1_APLANE = 128 * 32, S1_BPLANE = 256 * 32;        // bytes: one bf16 plane of the A / B tile
constexpr int S1_BUF = 3 * S1_APLANE + 3 * S1_BPLANE;            // 36 KB
constexpr int S1_LDS = 2 * S1_BUF;                               // 72 KB: two workgroups per CU
constexpr int S1_BIMG_TILE = 3 * S1_BPLANE;                      // one (column tile, K tile) of a weight image (gemm_split.hip)

__global__ __launch_bounds__(256, 2) void gemm_split128_kernel(const GemmBatch batch) {
    extern __shared__ __attribute__((aligned(16))) char s1_smem[];
    const GemmArgs& g = batch.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int step = __builtin_amdgcn_readfirstlane(g.step_ptr ? *g.step_ptr : g.step_imm);
    const int nbm = g.M / S1_BM, nbn = g.N / S1_BN;
    if ((int)blockIdx.x >= nbm * nbn) return;
    int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    if (g.xcd_rows > 0) {       // every XCD (private L2) a compact block of the tile grid; placement never changes results
        const int xr = g.xcd_rows, xc = 8 / xr, pr = nbm / xr, pc = nbn / xc;
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        bm = (xcd / xc) * pr + local / pc;
        bn = (xcd % xc) * pc + local % pc;
    }
    const int m0 = bm * S1_BM, n0 = bn * S1_BN;
    if (g.nact) {               // a tile without a live row is skipped (uniform over the workgroup, ahead of the first barrier)
        const int mlast = m0 + S1_BM - 1;
        const int l0 = m0 / g.nact_group, l1 = mlast / g.nact_group;
        int alive = 0;
        for (int l = l0 + lane; l <= l1; l += 64) alive |= g.nact[l] > (l == l0 ? m0 - l0 * g.nact_group : 0);
        if (!__any(alive)) return;
    }

    // ---- A rows: thread (r0, kc) stages floats [4 kc, 4 kc + 4) of rows r0 and r0 + 64 ----
    const int r0 = tid >> 2, kc = tid & 3;
    const Seg* const sgs[3] = {&g.a[0], &g.a[1], &g.a[2]};
    const int nseg = g.nseg;
    const float* abase[3]; long long ald[3]; int tiles[3], koff[3]; long long rowoff[3][2];
#pragma unroll
    for (int S = 0; S < 3; ++S) {
        const Seg& sg = *sgs[S];
        const bool act = nseg > S && !(sg.skip_first && step == 0 && !sg.first_base);
        const bool first = act && sg.first_base && step == 0;
        const bool gat = act && sg.rows && !first;
        abase[S] = !act ? nullptr : first ? sg.first_base : sg.base + (long long)(step * sg.step_mul + sg.step_add) * sg.slot_stride;
        ald[S] = sg.ld; tiles[S] = act ? sg.width / S1_BK : 0; koff[S] = sg.koff;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + r0 + 64 * i;
            rowoff[S][i] = (long long)(gat ? sg.rows[m] : m) * sg.ld + 4 * kc;
        }
    }
    const int c0 = tiles[0], c1 = tiles[0] + tiles[1], nt = tiles[0] + tiles[1] + tiles[2];
    // K tile t -> (segment, tile inside it): the A row pointers and the image tile of the weights
    auto a_ptr = [&](int t, int i) -> const float* {
        const int S = t < c0 ? 0 : t < c1 ? 1 : 2;
        const int tt = t - (S == 0 ? 0 : S == 1 ? c0 : c1);
        return abase[S] + rowoff[S][i] + (long long)tt * S1_BK;
    };
    auto b_tile = [&](int t) -> const char* {
        const int S = t < c0 ? 0 : t < c1 ? 1 : 2;
        const int tt = t - (S == 0 ? 0 : S == 1 ? c0 : c1);
        return reinterpret_cast<const char*>(g.Bimg) + ((long long)bn * (g.Ktot / S1_BK) + koff[S] / S1_BK + tt) * S1_BIMG_TILE;
    };
    struct GTile { f32x4 a[2]; };
    auto load_a = [&](GTile& gt, int t) {
        gt.a[0] = *reinterpret_cast<const f32x4*>(a_ptr(t, 0));
        gt.a[1] = *reinterpret_cast<const f32x4*>(a_ptr(t, 1));
    };
    // the B planes of K tile t -> LDS buffer `buf`: this wave's six 1-KB pieces of the 24-KB image tile
    auto dma_b = [&](int buf, int t) {
        const char* src = b_tile(t) + wave * 6144 + lane * 16;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)s1_smem)
                                 + (unsigned)(buf * S1_BUF + 3 * S1_APLANE) + (unsigned)__builtin_amdgcn_readfirstlane(wave * 6144) + (unsigned)(j * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src + j * 1024), "s"(dst) : "memory");
        }
    };
    auto split4 = [&](const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2) {       // gemm_split.hip's split, bit for bit
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 v = {x[2 * h], x[2 * h + 1]};
            const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            const f32x2 r1 = v - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
            const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
            const f32x2 r2 = r1 - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
            p0[h] = q0; p1[h] = q1; p2[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
        }
    };
    // (rows r0 and r0 + 64 share bit 4: one offset serves both)
    const int st_off = r0 * 32 + ((((kc >> 1) ^ (r0 >> 4)) & 1) * 16) + (kc & 1) * 8;
    auto store_a = [&](const GTile& gt, int buf) {
        char* base = s1_smem + buf * S1_BUF + st_off;
        u32x2 p0, p1, p2;
        split4(gt.a[0], p0, p1, p2);
        *reinterpret_cast<u32x2*>(base) = p0; *reinterpret_cast<u32x2*>(base + S1_APLANE) = p1; *reinterpret_cast<u32x2*>(base + 2 * S1_APLANE) = p2;
        split4(gt.a[1], p0, p1, p2);
        *reinterpret_cast<u32x2*>(base + 64 * 32) = p0; *reinterpret_cast<u32x2*>(base + S1_APLANE + 64 * 32) = p1;
        *reinterpret_cast<u32x2*>(base + 2 * S1_APLANE + 64 * 32) = p2;
    };
    const int fr_off = l31 * 32 + (((lh ^ (l31 >> 4)) & 1) * 16);
    auto frag_a = [&](int buf, int plane, int rb) {
        return *reinterpret_cast<const bf16x8*>(s1_smem + buf * S1_BUF + plane * S1_APLANE + (wm * 64 + rb * 32) * 32 + fr_off);
    };
    auto frag_b = [&](int buf, int plane, int c) {
        return *reinterpret_cast<const bf16x8*>(s1_smem + buf * S1_BUF + 3 * S1_APLANE + plane * S1_BPLANE + (wn * 128 + c * 32) * 32 + fr_off);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][c][r] = 0.0f;
    bf16x8 fb[4][3], fa[2][3];
#define CASV_S1_MMA(PA, PB)                                                                               \
    _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                                  \
            acc[rb_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb_][PA], fb[c_][PB], acc[rb_][c_], 0, 0, 0);

    GTile cur;
    if (nt > 0) {
        load_a(cur, 0);
        dma_b(0, 0);
        store_a(cur, 0);
        if (nt > 1) load_a(cur, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int c = 0; c < 4; ++c) fb[c][p] = frag_b(buf, p, c);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) fa[rb][p] = frag_a(buf, p, rb);
        }
        if (t + 1 < nt) dma_b(buf ^ 1, t + 1);          // (the other buffer's fragments were read before the last barrier)
        // the product order of gemm_split.hip
        CASV_S1_MMA(1, 1) CASV_S1_MMA(0, 2) CASV_S1_MMA(0, 1) CASV_S1_MMA(2, 0) CASV_S1_MMA(1, 0) CASV_S1_MMA(0, 0)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur.a[0]), "+v"(cur.a[1]) :: "memory");     // the transfers have landed; so have the rows of tile t + 1
        if (t + 1 < nt) store_a(cur, buf ^ 1);
        if (t + 2 < nt) load_a(cur, t + 2);
        __syncthreads();
    }
#undef CASV_S1_MMA

    // ---- epilogue: the LSTM cell of (row, unit), previous cell state straight from memory ----
    const Seg& sgc = g.c_in;
    const bool cfirst = sgc.first_base && step == 0;
    const bool czero = sgc.skip_first && step == 0 && !cfirst;
    const float* cin = cfirst ? sgc.first_base : sgc.base + (long long)(step * sgc.step_mul + sgc.step_add) * sgc.slot_stride;
    const bool cgat = sgc.rows && !cfirst;
    const int nb = n0 + wn * 128;                 // this wave's 128 columns: gates i, f, c~, o of 32 units
    const int u = nb / 4 + l31;                   // hidden unit of this lane
    float bi = 0.f, bf_ = 0.f, bg = 0.f, bo = 0.f;
    if (g.bias) { bi = g.bias[nb + l31]; bf_ = g.bias[nb + 32 + l31]; bg = g.bias[nb + 64 + l31]; bo = g.bias[nb + 96 + l31]; }
    float* cout = g.c_out.base + (long long)(step * g.c_out.step_mul + g.c_out.step_add) * g.c_out.slot_stride;
    float* hout = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        float cp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            cp[r] = czero ? 0.0f : cin[(long long)(cgat ? sgc.rows[m] : m) * sgc.ld + u];
        }
        float hv[16], cv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const LstmCellOut cell = lstm_cell(acc[rb][0][r] + bi, acc[rb][1][r] + bf_, acc[rb][2][r] + bg, acc[rb][3][r] + bo, cp[r]);
            hv[r] = cell.h; cv[r] = cell.c;
        }
        float* cb = cout + (long long)(m0 + wm * 64 + rb * 32 + 4 * lh) * g.c_out.ld + u;
        float* hb = hout + (long long)(m0 + wm * 64 + rb * 32 + 4 * lh) * g.out.ld + u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2);
            cb[(long long)dm * g.c_out.ld] = cv[r];
            hb[(long long)dm * g.out.ld] = hv[r];
        }
    }
}

// Jobs this shape takes (see the header); launch_gemm_split256 asks before it launches its own kernel.
bool gemm_split128_takes(int epi, const GemmBatch& b) {
    static const bool on = [] { const char* e = getenv("CASV_SPLIT128"); return e && e[0] == '1'; }();
    if (!on || epi != EPI_LSTM) return false;
    for (int j = 0; j < b.count; ++j) {
        const GemmArgs& g = b.g[j];
        if (g.epi_plain || !g.Bimg || g.M % S1_BM || g.N % S1_BN || !g.c_in.base || !g.c_out.base) return false;
        for (int i = 0; i < g.nseg; ++i) if (g.a[i].width % S1_BK || g.a[i].koff % S1_BK || g.a[i].ld % 4 || g.a[i].koff % 4) return false;
    }
    return true;
}

void launch_gemm_split128(const GemmBatch& b, hipStream_t stream) {
    static bool attr_set[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S1_LDS);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    int blocks = 0;
    GemmBatch bb = b;
    for (int j = 0; j < bb.count; ++j) {
        GemmArgs& g = bb.g[j];
        const int nbm = g.M / S1_BM, nbn = g.N / S1_BN;
        blocks = nbm * nbn > blocks ? nbm * nbn : blocks;
    }
    for (int j = 0; j < bb.count; ++j) {
        GemmArgs& g = bb.g[j];
        const int nbm = g.M / S1_BM, nbn = g.N / S1_BN;
        g.xcd_rows = 0;
        if ((nbm * nbn) % 8 != 0 || nbm * nbn != blocks) continue;
        double best = 0; int best_xr = 0;
        for (int xr = 1; xr <= 8; xr *= 2) {
            const int xc = 8 / xr;
            if (nbm % xr || nbn % xc) continue;
            const double cost = (double)g.M * xc + (double)g.N * xr;
            if (!best_xr || cost < best) { best = cost; best_xr = xr; }
        }
        g.xcd_rows = best_xr;
    }
    hipLaunchKernelGGL(gemm_split128_kernel, dim3(blocks, bb.count, 1), dim3(256), S1_LDS, stream, bb);
}

}  // namespace casv
