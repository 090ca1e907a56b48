// EXPERIMENT (option split_bf16 = 2, off by default; the fp32-input kernels of gemm.hip stay the library's arithmetic):
// the fused LSTM-cell GEMM on the bf16 matrix instruction with fp32-accurate sums, as 256x256 block tiles.
//
//   z = [x | ctx | h][M][K] . Wt[4U][K]^T + b ; (h', c') = cell(z, c)      or, for a job with the plain epilogue, C = A . Bt^T + bias
//
// Arithmetic (the same as gemm.hip's SPLIT variant): every fp32 operand value is taken apart into three bf16 values
// x = x0 + x1 + x2 (round to nearest; the remainders are exact in fp32, so the sum is exact) while its tile is staged into LDS, and
// a 16-deep K tile is contracted as six v_mfma_f32_32x32x16_bf16 products per 32x32 block with fp32 accumulation -- a1.b1, a0.b2,
// a0.b1, a2.b0, a1.b0, a0.b0; the dropped products a1.b2, a2.b1, a2.b2 are below 2^-25 |a||b| -- at 6/16 of the matrix-pipe time of
// the fp32-input instruction.  fp32-accurate, but not the k-ordered fmaf chain of the fp32 kernels: results agree to rounding.
//
// Why another tile shape: measured on the 128x128 variant (profiles/r04_split_bf16.txt), the matrix pipe is no longer what a K
// tile waits for -- per 24 products a wave issues ~100 vector instructions of splitting, 21 LDS and 4 memory instructions, and
// two such waves saturate a SIMD's issue.  Here 8 waves share a 256x256 tile, each wave 64 rows x 128 columns (all four gates
// of 32 units): 48 products per wave and K tile for the same 16 staged values per thread and 18 fragment reads -- half the
// instructions, half the L2 traffic per product.  One workgroup per CU (two waves per SIMD), 2 x 48 KB of tile buffers
// (three bf16 planes per operand, 32-byte rows whose 16-byte halves are swapped where bit 4 of the row is set: conflict-free
// ds_read_b128 without padding) + 64 KB for the previous cell state = the CU's whole LDS.
//
// Pipeline of tile t (fragment registers roll; one barrier per tile, behind the tile's first 24 products):
//   head:   plane 0 of B and of A of tile t <- LDS[t & 1]            (their registers are free since the end of tile t - 1)
//           products a1.b1, a0.b2, a0.b1
//   barrier (every wave has read all it needs of LDS[t & 1])
//           planes 1 and 2 of B of tile t + 1 <- LDS[(t + 1) & 1];  products a2.b0;  plane 2 of A of tile t + 1;
//           tile t + 2 (requested one tile ago) is split and stored into LDS[t & 1]; tile t + 3 is requested;
//           products a1.b0;  plane 1 of A of tile t + 1;  products a0.b0
// Tile loads are hidden from the compiler's wait bookkeeping (asm global_load + s_waitcnt, as in gemm.hip; checked by
// check_asm_loads.py): a tracked load pending across the loop's back edge would put a full wait at the loop head.
#include "common.h"
#include <cstdlib>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>

namespace casv {

#ifndef CASV_ABLM
#define CASV_ABLM 0         // timing-only builds (wrong results), a bit mask: 1 no split arithmetic, 2 no LDS stores, 4 no global loads, 8 no barrier, 16 no fragment reads
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Diagnostic build (-DCASV_S2_CLOCK): workgroup 0 stamps shader clock and wall clock around its K loop; s2_clock_dump() prints
// the shader clock the loop ran at (the bf16 matrix pipe with and without memory traffic beside it: power management).
#ifdef CASV_S2_CLOCK
__device__ unsigned long long g_s2_clk[4];
// ... and every workgroup adds the wall-clock ticks (10 ns) of its phases: [0] entry -> first steady tile (prologue), [1] steady loop,
// [2] tail tiles, [3] epilogue (cells, stores) -> end, [4] workgroups counted
__device__ unsigned long long g_s2_phase[8];
void s2_clock_dump() {
    unsigned long long h[4], ph[8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_s2_clk), sizeof(h));
    (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_s2_phase), sizeof(ph));
    if (h[1]) fprintf(stderr, "gemm_split256: loop of workgroup 0: %llu cycles in %llu wall ticks (10 ns) = %.3f GHz\n", h[0], h[1], (double)h[0] / (double)h[1] * 0.1);
    if (ph[4]) fprintf(stderr, "gemm_split256: per workgroup (average of %llu), us: prologue %.2f, steady loop %.2f, tail tiles %.2f, epilogue %.2f\n", ph[4],
                       ph[0] * 0.01 / ph[4], ph[1] * 0.01 / ph[4], ph[2] * 0.01 / ph[4], ph[3] * 0.01 / ph[4]);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_s2_phase), z, sizeof(z));
}
#endif
constexpr int S2_BM = 256, S2_BN = 256, S2_BK = 16;
constexpr int S2_PLANE = 256 * 32;                 // bytes: one bf16 plane of an operand tile
constexpr int S2_BUF = 6 * S2_PLANE;               // A planes 0..2, B planes 0..2
constexpr int S2_CELL = 8 * 64 * 128;              // per wave: 64 rows x 32 units of the previous cell state
#ifdef CASV_S2_CELL_LAST
constexpr int S2_TB = 0, S2_CB = 2 * S2_BUF;
#else
constexpr int S2_TB = S2_CELL, S2_CB = 0;
#endif
constexpr int S2_LDS = S2_CELL + 2 * S2_BUF;
constexpr int S2_BIMG_TILE = 3 * S2_PLANE;         // bytes of one (column tile, K tile) of a weight image: its three B planes as they lie in LDS       // [cell state | tile buffer 0 | tile buffer 1]

// BIMG: the B operand (weights that change only at casv_commit_weights) comes from an image in global memory that holds its
// tiles already split, plane by plane in the LDS layout (split_image_kernel below; the same values, so results do not depend on
// which form a launch takes): three 1-KB LDS-DMA transfers per wave and tile instead of two register loads, ~45 vector
// instructions and three LDS stores per thread -- the staging arithmetic of a tile halves.
template <int EPI, bool BIMG>
__global__ __launch_bounds__(512, 1) void gemm_split256_kernel(const GemmBatch batch) {
    extern __shared__ __attribute__((aligned(16))) char s2_smem[];
    const GemmArgs& g = batch.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CASV_S2_CLOCK
    const unsigned long long ph_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int step = __builtin_amdgcn_readfirstlane(g.step_ptr ? *g.step_ptr : g.step_imm);
    const int nbm = g.M / S2_BM, nbn = g.N / S2_BN;
    if ((int)blockIdx.x >= nbm * nbn) return;
    int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    if (g.xcd_rows > 0) {       // every XCD (private L2) a compact block of the tile grid; placement never changes results
        const int xr = g.xcd_rows, xc = 8 / xr, pr = nbm / xr, pc = nbn / xc;
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        bm = (xcd / xc) * pr + local / pc;
        bn = (xcd % xc) * pc + local % pc;
    }
    const int m0 = bm * S2_BM, n0 = bn * S2_BN;
    if (g.nact) {               // a tile without a live row is skipped (uniform over the workgroup, ahead of the first barrier)
        const int mlast = m0 + S2_BM - 1;
        const int l0 = m0 / g.nact_group, l1 = mlast / g.nact_group;
        int alive = 0;
        for (int l = l0 + lane; l <= l1; l += 64) alive |= g.nact[l] > (l == l0 ? m0 - l0 * g.nact_group : 0);
        if (!__any(alive)) return;
    }

    // ---- operand rows: thread (r0, kc) stages floats [4 kc, 4 kc + 4) of rows r0 and r0 + 128 of both operands ----
    const int r0 = tid >> 2, kc = tid & 3;
    const Seg sg0 = g.a[0], sg1 = g.a[1], sg2 = g.a[2], sgc = g.c_in;
    const Seg* const sgs[3] = {&sg0, &sg1, &sg2};
    const int nseg = g.nseg;
    const int* const no_rows = reinterpret_cast<const int*>(g.Bt);          // any readable word
    int ridx[3][2];
    bool act[3], gat[3];
#pragma unroll
    for (int S = 0; S < 3; ++S) {
        const Seg& sg = *sgs[S];
        act[S] = nseg > S && !(sg.skip_first && step == 0 && !sg.first_base);
        gat[S] = act[S] && sg.rows && !(sg.first_base && step == 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) ridx[S][i] = *(gat[S] ? sg.rows + (m0 + r0 + 128 * i) : no_rows);
    }
    const float* abase[3]; long long ald[3]; int tiles[3], koff[3];
#pragma unroll
    for (int S = 0; S < 3; ++S) {
        const Seg& sg = *sgs[S];
        const bool first = sg.first_base && step == 0;
        abase[S] = !act[S] ? nullptr : first ? sg.first_base : sg.base + (long long)(step * sg.step_mul + sg.step_add) * sg.slot_stride;
        ald[S] = sg.ld; tiles[S] = act[S] ? sg.width / S2_BK : 0; koff[S] = sg.koff;
#pragma unroll
        for (int i = 0; i < 2; ++i) if (!gat[S]) ridx[S][i] = m0 + r0 + 128 * i;
    }
    const int c0 = __builtin_amdgcn_readfirstlane(tiles[0]), c1 = __builtin_amdgcn_readfirstlane(tiles[0] + tiles[1]);
    const int nt = __builtin_amdgcn_readfirstlane(tiles[0] + tiles[1] + tiles[2]);
    auto arow = [&](int S, int i) { return reinterpret_cast<const char*>(abase[S] + (long long)ridx[S][i] * ald[S] + 4 * kc); };

    // running request pointers: one K tile further per request, re-based where the request stream enters the next segment
    int rseg = c0 > 0 ? 0 : c1 > c0 ? 1 : 2;
    int rleft = rseg == 0 ? c0 : rseg == 1 ? c1 - c0 : nt - c1;
    const char* ra0 = nullptr; const char* ra1 = nullptr;
    if (rseg == 0) { ra0 = arow(0, 0); ra1 = arow(0, 1); } else if (rseg == 1) { ra0 = arow(1, 0); ra1 = arow(1, 1); } else { ra0 = arow(2, 0); ra1 = arow(2, 1); }
    const float* rb0 = g.Bt + (long long)(n0 + r0) * g.Ktot + 4 * kc + koff[rseg];
    const float* rb1 = rb0 + (long long)128 * g.Ktot;
    // BIMG: this wave's 3 KB of the tile image of (column tile bn, K tile kt): the lane's 16 bytes of its first 1-KB piece
    const char* rbi = BIMG ? reinterpret_cast<const char*>(g.Bimg) + ((long long)bn * (g.Ktot / S2_BK) + koff[rseg] / S2_BK) * S2_BIMG_TILE + wave * 3072 + lane * 16 : nullptr;
    auto advance = [&]() {
        ra0 += S2_BK * 4; ra1 += S2_BK * 4; rb0 += S2_BK; rb1 += S2_BK;
        if (--rleft == 0) {
            asm volatile("" ::: "memory");
            if (rseg == 0 && c1 > c0) {
                rseg = 1; rleft = c1 - c0; ra0 = arow(1, 0); ra1 = arow(1, 1);
                rb0 += koff[1] - koff[0] - c0 * S2_BK; rb1 += koff[1] - koff[0] - c0 * S2_BK;
            } else if (rseg <= 1 && nt > c1) {
                const int kprev = rseg == 0 ? koff[0] + c0 * S2_BK : koff[1] + (c1 - c0) * S2_BK;
                rseg = 2; rleft = nt - c1; ra0 = arow(2, 0); ra1 = arow(2, 1);
                rb0 += koff[2] - kprev; rb1 += koff[2] - kprev;
            } else rleft = 1 << 30;
        }
    };
    // (the image pointer runs on its own: the planes of tile t + 2 are transferred a tile later than A's tile t + 3 is requested)
    int bseg = rseg, bleft = rleft;
    auto advance_b = [&]() {
        rbi += S2_BIMG_TILE;
        if (--bleft == 0) {
            if (bseg == 0 && c1 > c0) {
                bseg = 1; bleft = c1 - c0; rbi += (long long)((koff[1] - koff[0]) / S2_BK - c0) * S2_BIMG_TILE;
            } else if (bseg <= 1 && nt > c1) {
                const int kprev = bseg == 0 ? koff[0] + c0 * S2_BK : koff[1] + (c1 - c0) * S2_BK;
                bseg = 2; bleft = nt - c1; rbi += (long long)((koff[2] - kprev) / S2_BK) * S2_BIMG_TILE;
            } else bleft = 1 << 30;
        }
    };
    struct GTile { f32x4 a[2], b[2]; };
    // (one statement per load: the steady state spreads a tile's memory instructions over its products -- eight waves that
    // pass the barrier together and each issue five of them in a row queue up at the CU's one memory pipeline while the matrix
    // pipe has nothing to do: 3 960 instead of 3 210 cycles per tile, profiles/r04_split_bf16.txt)
    auto request_a0 = [&](GTile& gt) { if (!(CASV_ABLM & 4)) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.a[0]) : "v"(ra0)); };
    auto request_a1 = [&](GTile& gt) { if (!(CASV_ABLM & 4)) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.a[1]) : "v"(ra1)); };
    auto request_b0 = [&](GTile& gt) { if (!(CASV_ABLM & 4) && !BIMG) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.b[0]) : "v"(rb0)); };
    auto request_b1 = [&](GTile& gt) { if (!(CASV_ABLM & 4) && !BIMG) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.b[1]) : "v"(rb1)); };
    auto request = [&](GTile& gt) { request_a0(gt); request_a1(gt); request_b0(gt); request_b1(gt); };
#define CASV_S2_LANDED(G) { if (!(CASV_ABLM & (4 | 32))) { if (BIMG) asm volatile("s_waitcnt vmcnt(0)" : "+v"(G.a[0]), "+v"(G.a[1])); \
                            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(G.a[0]), "+v"(G.a[1]), "+v"(G.b[0]), "+v"(G.b[1])); } }
    // BIMG: the B planes of the tile the running image pointer stands at -> LDS buffer `buf`, this wave's three 1-KB pieces
    auto dma_b1 = [&](int buf, int j) {
        if (CASV_ABLM & 4) return;
        {
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)s2_smem)
                                 + (unsigned)(S2_TB + buf * S2_BUF + 3 * S2_PLANE) + (unsigned)__builtin_amdgcn_readfirstlane(wave * 3072) + (unsigned)(j * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(rbi + j * 1024), "s"(dst) : "memory");
        }
    };
    auto dma_b = [&](int buf) { dma_b1(buf, 0); dma_b1(buf, 1); dma_b1(buf, 2); };

    // ---- staging: split and store ----
    auto split4 = [&](const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (CASV_ABLM & 1) { p0[h] = __float_as_uint(x[2 * h]); p1[h] = __float_as_uint(x[2 * h + 1]); p2[h] = p0[h] ^ p1[h]; continue; }
            const f32x2 v = {x[2 * h], x[2 * h + 1]};
            const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            const f32x2 r1 = v - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
            const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
            const f32x2 r2 = r1 - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
            p0[h] = q0; p1[h] = q1; p2[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
        }
    };
    // (rows r0 and r0 + 128 share bit 4: one offset serves both)
    const int st_off = r0 * 32 + ((((kc >> 1) ^ (r0 >> 4)) & 1) * 16) + (kc & 1) * 8;
    auto store_op = [&](const f32x4 v0, const f32x4 v1, int buf, int plane0) {
        if (CASV_ABLM & 2) { asm volatile("" :: "v"(v0), "v"(v1)); return; }
        char* base = s2_smem + S2_TB + buf * S2_BUF + plane0 * S2_PLANE + st_off;
        u32x2 p0, p1, p2;
        split4(v0, p0, p1, p2);
        *reinterpret_cast<u32x2*>(base) = p0; *reinterpret_cast<u32x2*>(base + S2_PLANE) = p1; *reinterpret_cast<u32x2*>(base + 2 * S2_PLANE) = p2;
        split4(v1, p0, p1, p2);
        *reinterpret_cast<u32x2*>(base + 128 * 32) = p0; *reinterpret_cast<u32x2*>(base + S2_PLANE + 128 * 32) = p1;
        *reinterpret_cast<u32x2*>(base + 2 * S2_PLANE + 128 * 32) = p2;
    };
    // the split in two halves per value pair (h = 0, 1 of a staged f32x4), so that each half fits behind one product
    auto split_l1 = [&](const f32x4 x, int h, unsigned& q0, f32x2& r1) {
        const f32x2 v = {x[2 * h], x[2 * h + 1]};
        if (CASV_ABLM & 1) { q0 = __float_as_uint(v[0]); r1 = v; return; }
        q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        r1 = v - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};       // (two v_add_f32: v_pk_add_f32 measured slower)
    };
    auto split_l23 = [&](const unsigned q0, const f32x2 r1, int h, u32x2& p0, u32x2& p1, u32x2& p2) {
        if (CASV_ABLM & 1) { p0[h] = q0; p1[h] = __float_as_uint(r1[0]); p2[h] = __float_as_uint(r1[1]); return; }
        const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
        const f32x2 r2 = r1 - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
        p0[h] = q0; p1[h] = q1; p2[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
    };
    auto store_row = [&](const u32x2 p0, const u32x2 p1, const u32x2 p2, int buf, int plane0, int i) {      // row r0 + 128 i
        if (CASV_ABLM & 2) { asm volatile("" :: "v"(p0), "v"(p1), "v"(p2)); return; }
        char* base = s2_smem + S2_TB + buf * S2_BUF + plane0 * S2_PLANE + st_off + i * 128 * 32;
        *reinterpret_cast<u32x2*>(base) = p0; *reinterpret_cast<u32x2*>(base + S2_PLANE) = p1; *reinterpret_cast<u32x2*>(base + 2 * S2_PLANE) = p2;
    };
    // a lane's 8 k of its row: the 16-byte half lh (k = 8 lh .. 8 lh + 7, the same for both operands)
    const int fr_off = l31 * 32 + (((lh ^ (l31 >> 4)) & 1) * 16);
    auto frag_a = [&](int buf, int plane, int rb) {
        if (CASV_ABLM & 16) { bf16x8 z; asm volatile("" : "=v"(z)); return z; }
        return *reinterpret_cast<const bf16x8*>(s2_smem + S2_TB + buf * S2_BUF + plane * S2_PLANE + (wm * 64 + rb * 32) * 32 + fr_off);
    };
    auto frag_b = [&](int buf, int plane, int c) {
        if (CASV_ABLM & 16) { bf16x8 z; asm volatile("" : "=v"(z)); return z; }
        return *reinterpret_cast<const bf16x8*>(s2_smem + S2_TB + buf * S2_BUF + (3 + plane) * S2_PLANE + (wn * 128 + c * 32) * 32 + fr_off);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][c][r] = 0.0f;
    bf16x8 fb[4][3], fa[2][3];
    // (timing-only build -DCASV_S2_SHAPE16, wrong results: every product as two v_mfma_f32_16x16x32_bf16 -- the same matrix-pipe cycles
    // and operand registers -- to see what clock the real loop holds on that instruction shape: profiles/r05_mfma_shape_probe.txt)
#ifdef CASV_S2_SHAPE16
#define CASV_S2_PROD(A, B, ACC) { \
        f32x4 p0_ = __builtin_shufflevector(ACC, ACC, 0, 1, 2, 3), p1_ = __builtin_shufflevector(ACC, ACC, 4, 5, 6, 7);                       \
        p0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, p0_, 0, 0, 0); p1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, p1_, 0, 0, 0);   \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { ACC[e_] = p0_[e_]; ACC[4 + e_] = p1_[e_]; } }
#else
#define CASV_S2_PROD(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0);
#endif
#define CASV_S2_MMA(PA, PB)                                                                               \
    _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                                  \
            CASV_S2_PROD(fa[rb_][PA], fb[c_][PB], acc[rb_][c_])
    // scheduling fence that only vector arithmetic may cross (the split's arithmetic finds its own place between the products;
    // matrix, LDS and memory instructions stay where the pipeline above puts them)
#define CASV_S2_PIN __builtin_amdgcn_sched_barrier(0x2);

    // ---- prologue: tiles 0 and 1 into LDS, tile 2 requested; the fragments a tile expects in registers ----
    GTile gt;
    if (nt > 1) {       // tiles 0 and 1 requested together: one memory round trip in front of the first product instead of two
        GTile g1;
        request(gt); if (BIMG) { dma_b(0); advance_b(); } advance();
        request(g1); if (BIMG) { dma_b(1); advance_b(); } advance();
        CASV_S2_LANDED(gt); CASV_S2_LANDED(g1);
        store_op(gt.a[0], gt.a[1], 0, 0); if (!BIMG) store_op(gt.b[0], gt.b[1], 0, 3);
        store_op(g1.a[0], g1.a[1], 1, 0); if (!BIMG) store_op(g1.b[0], g1.b[1], 1, 3);
    } else if (nt > 0) {
        request(gt); if (BIMG) { dma_b(0); advance_b(); } advance(); CASV_S2_LANDED(gt);
        store_op(gt.a[0], gt.a[1], 0, 0); if (!BIMG) store_op(gt.b[0], gt.b[1], 0, 3);
    }
    if (nt > 2) { request(gt); advance(); }
    __syncthreads();
    if (nt > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { fb[c][1] = frag_b(0, 1, c); fb[c][2] = frag_b(0, 2, c); }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) { fa[rb][1] = frag_a(0, 1, rb); fa[rb][2] = frag_a(0, 2, rb); }
    }

    // steady state: tiles T+1..T+3 exist, no conditionals.  Behind the barrier the issue order is written out product by product
    // with a full scheduling fence behind each: a product, at most one fragment read, half of a value pair's split (5-6 vector
    // instructions), the three 8-byte LDS stores of a row once its four values are split -- so that the matrix pipe is fed every
    // ~32 cycles by this wave alone.  Left to itself the compiler issues the ~45 vector instructions of an operand's split in
    // one run (180 cycles without a product), and its group-barrier solver gives up on all but the first block of such a tile.
#define CASV_S2_M1(RB, C, PA, PB) CASV_S2_PROD(fa[RB][PA], fb[C][PB], acc[RB][C])
#define CASV_S2_FENCE __builtin_amdgcn_sched_barrier(0);
#define CASV_S2_TILE(T, PAR) \
    { \
        u32x2 w0_, w1_, w2_; unsigned q0_; f32x2 r1_;                                                          \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) fb[c_][0] = frag_b(PAR, 0, c_);                       \
        _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) fa[rb_][0] = frag_a(PAR, 0, rb_);                  \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_MMA(1, 1) CASV_S2_MMA(0, 2) CASV_S2_MMA(0, 1)                                                  \
        CASV_S2_FENCE                                                                                          \
        if (!(CASV_ABLM & 8)) __syncthreads();                                                                 \
        CASV_S2_LANDED(gt);                                                                                    \
        CASV_S2_M1(0, 0, 2, 0)                                                                                 \
        fb[0][1] = frag_b(1 - PAR, 1, 0);                                                                      \
        split_l1(gt.a[0], 0, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 1, 2, 0)                                                                                 \
        fb[1][1] = frag_b(1 - PAR, 1, 1);                                                                      \
        split_l23(q0_, r1_, 0, w0_, w1_, w2_);                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 2, 2, 0)                                                                                 \
        fb[2][1] = frag_b(1 - PAR, 1, 2);                                                                      \
        split_l1(gt.a[0], 1, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 3, 2, 0)                                                                                 \
        fb[3][1] = frag_b(1 - PAR, 1, 3);                                                                      \
        split_l23(q0_, r1_, 1, w0_, w1_, w2_);                                                                 \
        store_row(w0_, w1_, w2_, PAR, 0, 0);                                                                   \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 0, 2, 0)                                                                                 \
        fb[0][2] = frag_b(1 - PAR, 2, 0);                                                                      \
        split_l1(gt.a[1], 0, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 1, 2, 0)                                                                                 \
        fb[1][2] = frag_b(1 - PAR, 2, 1);                                                                      \
        split_l23(q0_, r1_, 0, w0_, w1_, w2_);                                                                 \
        request_a0(gt);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 2, 2, 0)                                                                                 \
        fb[2][2] = frag_b(1 - PAR, 2, 2);                                                                      \
        split_l1(gt.a[1], 1, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 3, 2, 0)                                                                                 \
        fb[3][2] = frag_b(1 - PAR, 2, 3);                                                                      \
        split_l23(q0_, r1_, 1, w0_, w1_, w2_);                                                                 \
        store_row(w0_, w1_, w2_, PAR, 0, 1);                                                                   \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 0, 1, 0)                                                                                 \
        fa[0][2] = frag_a(1 - PAR, 2, 0);                                                                      \
        split_l1(gt.b[0], 0, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 1, 1, 0)                                                                                 \
        fa[1][2] = frag_a(1 - PAR, 2, 1);                                                                      \
        split_l23(q0_, r1_, 0, w0_, w1_, w2_);                                                                 \
        request_a1(gt);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 2, 1, 0)                                                                                 \
        split_l1(gt.b[0], 1, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 3, 1, 0)                                                                                 \
        split_l23(q0_, r1_, 1, w0_, w1_, w2_);                                                                 \
        store_row(w0_, w1_, w2_, PAR, 3, 0);                                                                   \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 0, 1, 0)                                                                                 \
        split_l1(gt.b[1], 0, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 1, 1, 0)                                                                                 \
        split_l23(q0_, r1_, 0, w0_, w1_, w2_);                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 2, 1, 0)                                                                                 \
        split_l1(gt.b[1], 1, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 3, 1, 0)                                                                                 \
        split_l23(q0_, r1_, 1, w0_, w1_, w2_);                                                                 \
        store_row(w0_, w1_, w2_, PAR, 3, 1);                                                                   \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 0, 0, 0)                                                                                 \
        fa[0][1] = frag_a(1 - PAR, 1, 0);                                                                      \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 1, 0, 0)                                                                                 \
        fa[1][1] = frag_a(1 - PAR, 1, 1);                                                                      \
        request_b0(gt);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 2, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 3, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 0, 0, 0)                                                                                 \
        request_b1(gt);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 1, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 2, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 3, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        advance();                                                                                             \
    }
    // the same with the B planes from the weight image (BIMG): only A is staged through registers
#define CASV_S2_TILE_BI(T, PAR) \
    { \
        u32x2 w0_, w1_, w2_; unsigned q0_; f32x2 r1_;                                                          \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) fb[c_][0] = frag_b(PAR, 0, c_);                       \
        _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) fa[rb_][0] = frag_a(PAR, 0, rb_);                  \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_MMA(1, 1) CASV_S2_MMA(0, 2) CASV_S2_MMA(0, 1)                                                  \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_LANDED(gt);   /* (this wave's plane transfers of the previous tile too: they are read behind the barrier) */ \
        if (!(CASV_ABLM & 8)) __syncthreads();                                                                 \
        CASV_S2_M1(0, 0, 2, 0)                                                                                 \
        fb[0][1] = frag_b(1 - PAR, 1, 0);                                                                      \
        split_l1(gt.a[0], 0, q0_, r1_);                                                                        \
        dma_b1(PAR, 0);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 1, 2, 0)                                                                                 \
        fb[1][1] = frag_b(1 - PAR, 1, 1);                                                                      \
        split_l23(q0_, r1_, 0, w0_, w1_, w2_);                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 2, 2, 0)                                                                                 \
        fb[2][1] = frag_b(1 - PAR, 1, 2);                                                                      \
        split_l1(gt.a[0], 1, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 3, 2, 0)                                                                                 \
        fb[3][1] = frag_b(1 - PAR, 1, 3);                                                                      \
        split_l23(q0_, r1_, 1, w0_, w1_, w2_);                                                                 \
        store_row(w0_, w1_, w2_, PAR, 0, 0);                                                                   \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 0, 2, 0)                                                                                 \
        fb[0][2] = frag_b(1 - PAR, 2, 0);                                                                      \
        split_l1(gt.a[1], 0, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 1, 2, 0)                                                                                 \
        fb[1][2] = frag_b(1 - PAR, 2, 1);                                                                      \
        split_l23(q0_, r1_, 0, w0_, w1_, w2_);                                                                 \
        request_a0(gt);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 2, 2, 0)                                                                                 \
        fb[2][2] = frag_b(1 - PAR, 2, 2);                                                                      \
        split_l1(gt.a[1], 1, q0_, r1_);                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 3, 2, 0)                                                                                 \
        fb[3][2] = frag_b(1 - PAR, 2, 3);                                                                      \
        split_l23(q0_, r1_, 1, w0_, w1_, w2_);                                                                 \
        store_row(w0_, w1_, w2_, PAR, 0, 1);                                                                   \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 0, 1, 0)                                                                                 \
        fa[0][2] = frag_a(1 - PAR, 2, 0);                                                                      \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 1, 1, 0)                                                                                 \
        fa[1][2] = frag_a(1 - PAR, 2, 1);                                                                      \
        request_a1(gt);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 2, 1, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 3, 1, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 0, 1, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 1, 1, 0)                                                                                 \
        dma_b1(PAR, 1);                                                                                        \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 2, 1, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 3, 1, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 0, 0, 0)                                                                                 \
        fa[0][1] = frag_a(1 - PAR, 1, 0);                                                                      \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 1, 0, 0)                                                                                 \
        fa[1][1] = frag_a(1 - PAR, 1, 1);                                                                      \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 2, 0, 0)                                                                                 \
        dma_b1(PAR, 2); advance_b();                                                                           \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(0, 3, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 0, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 1, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 2, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        CASV_S2_M1(1, 3, 0, 0)                                                                                 \
        CASV_S2_FENCE                                                                                          \
        advance();                                                                                             \
    }
    // the last tiles: the same order with (workgroup-uniform) conditions; staging in one piece behind the barrier
#define CASV_S2_TAIL(T, PAR)                                                                              \
    {                                                                                                     \
        const bool next_ = (T) + 1 < nt;                                                                  \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) fb[c_][0] = frag_b(PAR, 0, c_);                  \
        _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) fa[rb_][0] = frag_a(PAR, 0, rb_);             \
        CASV_S2_MMA(1, 1) CASV_S2_MMA(0, 2) CASV_S2_MMA(0, 1)                                             \
        CASV_S2_LANDED(gt);     /* (unconditional: no path carries a request past a tile, whatever the checker assumes about the conditions; BIMG: the previous tile's plane transfers) */ \
        __syncthreads();                                                                                  \
        if (BIMG && (T) + 2 < nt) { dma_b(PAR); advance_b(); }                                            \
        if (next_) { _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) { fb[c_][1] = frag_b(1 - PAR, 1, c_); fb[c_][2] = frag_b(1 - PAR, 2, c_); } }   \
        if ((T) + 2 < nt) {                                                                               \
            store_op(gt.a[0], gt.a[1], PAR, 0); if (!BIMG) store_op(gt.b[0], gt.b[1], PAR, 3);            \
            if ((T) + 3 < nt) { request(gt); advance(); }                                                 \
        }                                                                                                 \
        CASV_S2_MMA(2, 0)                                                                                 \
        if (next_) { _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) fa[rb_][2] = frag_a(1 - PAR, 2, rb_); }   \
        CASV_S2_MMA(1, 0)                                                                                 \
        if (next_) { _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) fa[rb_][1] = frag_a(1 - PAR, 1, rb_); }   \
        CASV_S2_MMA(0, 0)                                                                                 \
    }
    int t = 0;
#ifdef CASV_S2_CLOCK
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), cr0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (BIMG) {
        for (; t + 4 < nt; t += 2) {
            CASV_S2_TILE_BI(t, 0)
            CASV_S2_TILE_BI(t + 1, 1)
        }
    } else
    for (; t + 4 < nt; t += 2) {
        CASV_S2_TILE(t, 0)
        CASV_S2_TILE(t + 1, 1)
    }

#ifdef CASV_S2_CLOCK
    const unsigned long long ph_t2 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        g_s2_clk[0] = __builtin_amdgcn_s_memtime() - ck0; g_s2_clk[1] = ph_t2 - cr0;
    }
#endif
    // ---- previous cell state: every wave fetches the 64 rows x 32 units it will need itself, as LDS-DMA under the last tiles ----
    const bool plain = EPI == EPI_PLAIN || g.epi_plain;
    const bool cfirst = !plain && sgc.first_base && step == 0;
    const bool czero = !plain && sgc.skip_first && step == 0 && !cfirst;
    char* const cellw = s2_smem + S2_CB + wave * (64 * 128);        // (LDS-DMA destinations beyond 64 KB work too: the -DCASV_S2_CELL_LAST build, cell state at 96..160 KB, passes the same tests)
    if (!plain && !czero) {
        const float* cin = cfirst ? sgc.first_base : sgc.base + (long long)(step * sgc.step_mul + sgc.step_add) * sgc.slot_stride;
        const bool cgat = sgc.rows && !cfirst;
        int crow[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int m = m0 + wm * 64 + j * 8 + (lane >> 3); crow[j] = cgat ? sgc.rows[m] : m; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* src = cin + (long long)crow[j] * sgc.ld + bn * 64 + wn * 32 + 4 * (lane & 7);
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)cellw + (unsigned)(j * 1024));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    }
    for (; t + 1 < nt; t += 2) {
        CASV_S2_TAIL(t, 0)
        CASV_S2_TAIL(t + 1, 1)
    }
    if (t < nt) CASV_S2_TAIL(t, 0)
#undef CASV_S2_TILE
#undef CASV_S2_TILE_BI
#undef CASV_S2_TAIL
#undef CASV_S2_M1
#undef CASV_S2_FENCE
#undef CASV_S2_MMA
#undef CASV_S2_PROD
#undef CASV_S2_PIN
    CASV_S2_LANDED(gt);                                  // (pins the staging registers until nothing can be in flight into them)
#ifdef CASV_S2_CLOCK
    const unsigned long long ph_t3 = __builtin_amdgcn_s_memrealtime();
#endif
#undef CASV_S2_LANDED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the wave's own cell-state transfers (an LDS-DMA must not outlive its workgroup either)

    // ---- epilogue ----
    if (plain) {
        float* cbase = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        float bcol[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bcol[c] = g.bias ? g.bias[n0 + wn * 128 + c * 32 + l31] : 0.0f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float* cb = cbase + (long long)(m0 + wm * 64 + rb * 32 + 4 * lh) * g.out.ld + n0 + wn * 128 + l31;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) cb[(long long)((r & 3) + 8 * (r >> 2)) * g.out.ld + c * 32] = acc[rb][c][r] + bcol[c];
        }
        return;
    }
    const int nb = n0 + wn * 128;                 // this wave's 128 columns: gates i, f, c~, o of 32 units
    const int u = nb / 4 + l31;                   // hidden unit of this lane
    float bi = 0.f, bf_ = 0.f, bg = 0.f, bo = 0.f;
    if (g.bias) { bi = g.bias[nb + l31]; bf_ = g.bias[nb + 32 + l31]; bg = g.bias[nb + 64 + l31]; bo = g.bias[nb + 96 + l31]; }
    float* cout = g.c_out.base + (long long)(step * g.c_out.step_mul + g.c_out.step_add) * g.c_out.slot_stride;
    float* hout = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        float hv[16], cv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float cprev = czero ? 0.0f : *reinterpret_cast<const float*>(cellw + row * 128 + l31 * 4);
            const LstmCellOut cell = lstm_cell(acc[rb][0][r] + bi, acc[rb][1][r] + bf_, acc[rb][2][r] + bg, acc[rb][3][r] + bo, cprev);
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
#ifdef CASV_S2_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the stores have left)
    if (tid == 0) {
        const unsigned long long ph_t4 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_s2_phase[0], cr0 - ph_t0); atomicAdd(&g_s2_phase[1], ph_t2 - cr0); atomicAdd(&g_s2_phase[2], ph_t3 - ph_t2);
        atomicAdd(&g_s2_phase[3], ph_t4 - ph_t3); atomicAdd(&g_s2_phase[4], 1ull);
    }
#endif
}

// ---- weight images (BIMG) ----
// One thread per four k of a row of Bt [N][K]: the same split as the staging path (bit for bit), written where that path's LDS
// stores would put it -- tile (n / 256, k / 16), plane, row n % 256, the 16-byte halves swapped where bit 4 of the row is set.
__global__ void split_image_kernel(const float* __restrict__ Bt, int N, int K, char* __restrict__ img) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int kq4 = K / 4;
    if (i >= (long long)N * kq4) return;
    const int n = (int)(i / kq4), k = 4 * (int)(i % kq4);
    const f32x4 x = *reinterpret_cast<const f32x4*>(Bt + (long long)n * K + k);
    u32x2 p[3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 v = {x[2 * h], x[2 * h + 1]};
        const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        const f32x2 r1 = v - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
        const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
        const f32x2 r2 = r1 - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
        p[0][h] = q0; p[1][h] = q1; p[2][h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
    }
    const int r = n % S2_BN, kc = (k % S2_BK) / 4;
    char* tile = img + ((long long)(n / S2_BN) * (K / S2_BK) + k / S2_BK) * S2_BIMG_TILE;
    const int off = r * 32 + ((((kc >> 1) ^ (r >> 4)) & 1) * 16) + (kc & 1) * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(tile + q * S2_PLANE + off) = p[q];
}

// Images are made on first use (on the launch's stream, ahead of the launch) and kept per (weight pointer, shape) until
// gemm_split_invalidate(Bt): called wherever a weight buffer changes or goes away (casv_commit_weights, casv_model_destroy) --
// per buffer, so that another handle decoding on another thread keeps the images it is using.
struct SplitImage { void* img; int N, K; };
static std::mutex g_img_mutex;
static std::map<std::pair<int, const float*>, SplitImage> g_images;     // key: (device, Bt)
static const void* split_image_of(const float* Bt, int N, int K, hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_img_mutex);
    auto it = g_images.find({dev, Bt});
    if (it != g_images.end() && it->second.N == N && it->second.K == K) return it->second.img;
    // (no allocation while the stream records a graph: that launch stages B itself -- same values -- and the image is made by
    // the first eager launch that wants it)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cap != hipStreamCaptureStatusNone) return nullptr;
    if (it != g_images.end()) {
        (void)hipFree(it->second.img);
        g_images.erase(it);
        gemm_split_bump_epoch();
    }
    void* img = nullptr;
    if (hipMalloc(&img, (size_t)N * K * 6) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const long long n4 = (long long)N * (K / 4);
    hipLaunchKernelGGL(split_image_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, Bt, N, K, reinterpret_cast<char*>(img));
    g_images[{dev, Bt}] = SplitImage{img, N, K};
    return img;
}
static bool split256_set_attributes() {        // the dynamic-LDS size of the four variants, once per device (not inside a recording)
    static bool attr_set[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && attr_set[dev]) return true;
    const void* fns[4] = {reinterpret_cast<const void*>(&gemm_split256_kernel<EPI_PLAIN, false>), reinterpret_cast<const void*>(&gemm_split256_kernel<EPI_LSTM, false>),
                          reinterpret_cast<const void*>(&gemm_split256_kernel<EPI_PLAIN, true>), reinterpret_cast<const void*>(&gemm_split256_kernel<EPI_LSTM, true>)};
    for (const void* fn : fns)
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
    return true;
}
void gemm_split_prepare(const float* Bt, int N, int K, hipStream_t stream) {
    (void)split256_set_attributes();
    if (Bt && N > 0 && N % S2_BN == 0 && K > 0 && K % S2_BK == 0) (void)split_image_of(Bt, N, K, stream);
}
void gemm_split_invalidate(const float* Bt) {        // the image(s) of one weight buffer; nullptr: all
    std::lock_guard<std::mutex> lock(g_img_mutex);
    for (auto it = g_images.begin(); it != g_images.end();) {
        if (!Bt || it->first.second == Bt) { (void)hipFree(it->second.img); it = g_images.erase(it); gemm_split_bump_epoch(); }     // (hipFree waits for the device: nothing still reads it; a captured step graph that holds the address is rebuilt)
        else ++it;
    }
}

// Which jobs of a launch can go as 256x256 tiles: whole tiles only, inference outputs only (no gate / second-h / precomputed-term
// side channels of the train step), K segments in whole tiles.
static bool split256_job_ok(int epi, const GemmArgs& g) {
    if (g.M <= 0 || g.M % S2_BM || g.N % S2_BN || g.nseg < 1) return false;
    if (g.accumulate || g.ksplit > 1 || g.zinit.base || g.gates_out.base || g.out2.base) return false;
    for (int i = 0; i < g.nseg; ++i) if (g.a[i].width % S2_BK || g.a[i].koff % 4 || g.a[i].ld % 4) return false;
    if (g.Ktot % 4) return false;
    if (epi == EPI_LSTM && !g.epi_plain && (!g.c_out.base || !g.c_in.base)) return false;
    return true;
}

bool gemm_split256_wants(int epi, const GemmArgs& g) {
    return split256_job_ok(epi, g) && (g.M / S2_BM) * (g.N / S2_BN) >= 128;      // fewer tiles: the smaller tile shapes do better
}

// true: launched.  false: not eligible as a whole (the caller takes the 128x128 path).
bool launch_gemm_split256(int epi, const GemmBatch& b, hipStream_t stream) {
    int blocks = 0;
    GemmBatch bb = b;
    for (int j = 0; j < b.count; ++j) {
        if (!split256_job_ok(epi, b.g[j])) return false;
        GemmArgs& g = bb.g[j];
        const int nbm = g.M / S2_BM, nbn = g.N / S2_BN;
        blocks = nbm * nbn > blocks ? nbm * nbn : blocks;
    }
    for (int j = 0; j < b.count; ++j) {
        GemmArgs& g = bb.g[j];
        const int nbm = g.M / S2_BM, nbn = g.N / S2_BN;
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
    // weight images: every job of the launch must have one (static weights, K in whole tiles)
    static const bool images_off = [] { const char* e = getenv("CASV_SPLIT_IMAGES"); return e && e[0] == '0'; }();
    bool bimg = !images_off;
    for (int j = 0; j < bb.count && bimg; ++j) bimg = bb.g[j].b_static && bb.g[j].Ktot % S2_BK == 0;
    for (int j = 0; j < bb.count && bimg; ++j) {
        for (int i = 0; i < bb.g[j].nseg; ++i) if (bb.g[j].a[i].koff % S2_BK) bimg = false;
        if (bimg) { bb.g[j].Bimg = split_image_of(bb.g[j].Bt, bb.g[j].N, bb.g[j].Ktot, stream); bimg = bb.g[j].Bimg != nullptr; }
    }
    if (!split256_set_attributes()) return false;
    const int e = (epi == EPI_LSTM ? 1 : 0) + (bimg ? 2 : 0);
    const dim3 grid(blocks, bb.count, 1);
    if (e == 3) hipLaunchKernelGGL((gemm_split256_kernel<EPI_LSTM, true>), grid, dim3(512), S2_LDS, stream, bb);
    else if (e == 2) hipLaunchKernelGGL((gemm_split256_kernel<EPI_PLAIN, true>), grid, dim3(512), S2_LDS, stream, bb);
    else if (e == 1) hipLaunchKernelGGL((gemm_split256_kernel<EPI_LSTM, false>), grid, dim3(512), S2_LDS, stream, bb);
    else hipLaunchKernelGGL((gemm_split256_kernel<EPI_PLAIN, false>), grid, dim3(512), S2_LDS, stream, bb);
    return true;
}

}  // namespace casv
