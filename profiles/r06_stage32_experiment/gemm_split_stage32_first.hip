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
    bool act[3], gat[3];
#pragma unroll
    for (int S = 0; S < 3; ++S) {
        const Seg& sg = *sgs[S];
        act[S] = nseg > S && !(sg.skip_first && step == 0 && !sg.first_base);
        gat[S] = act[S] && sg.rows && !(sg.first_base && step == 0);
    }
    const float* abase[3]; long long ald[3]; int tiles[3], koff[3];
#pragma unroll
    for (int S = 0; S < 3; ++S) {
        const Seg& sg = *sgs[S];
        const bool first = sg.first_base && step == 0;
        abase[S] = !act[S] ? nullptr : first ? sg.first_base : sg.base + (long long)(step * sg.step_mul + sg.step_add) * sg.slot_stride;
        ald[S] = sg.ld; tiles[S] = act[S] ? sg.width / S2_BK : 0; koff[S] = sg.koff;
    }
    const int c0 = __builtin_amdgcn_readfirstlane(tiles[0]), c1 = __builtin_amdgcn_readfirstlane(tiles[0] + tiles[1]);
    const int nt = __builtin_amdgcn_readfirstlane(tiles[0] + tiles[1] + tiles[2]);
    // (a gathered segment's row indices are fetched where the request stream enters it -- twice per K loop at most -- instead of
    // waiting in six registers)
    auto arow = [&](int S, int i) {
        const int m = m0 + r0 + 128 * i;
        const int r = gat[S] ? sgs[S]->rows[m] : m;
        return reinterpret_cast<const char*>(abase[S] + (long long)r * ald[S] + 4 * kc);
    };

    // running request pointers: one K tile further per request, re-based where the request stream enters the next segment
    int rseg = c0 > 0 ? 0 : c1 > c0 ? 1 : 2;
    int rleft = rseg == 0 ? c0 : rseg == 1 ? c1 - c0 : nt - c1;
    const char* ra0 = nullptr; const char* ra1 = nullptr;
    if (rseg == 0) { ra0 = arow(0, 0); ra1 = arow(0, 1); } else if (rseg == 1) { ra0 = arow(1, 0); ra1 = arow(1, 1); } else { ra0 = arow(2, 0); ra1 = arow(2, 1); }
    const float* rb0 = g.Bt + (long long)(n0 + r0) * g.Ktot + 4 * kc + koff[rseg];
    const float* rb1 = rb0 + (long long)128 * g.Ktot;
    // BIMG: the tile image of (column tile bn, K tile kt) -- a workgroup-uniform address (scalar registers); the lane's 16 bytes of this
    // wave's 1-KB piece of a plane sit at img_lane behind it
    const char* rbi = nullptr;
    if (BIMG) {
        const unsigned long long a64 = (unsigned long long)(reinterpret_cast<const char*>(g.Bimg) + ((long long)bn * (g.Ktot / S2_BK) + koff[rseg] / S2_BK) * S2_BIMG_TILE);
        rbi = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a64 >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a64));
    }
    const unsigned img_lane = wave * 1024 + lane * 16;
    auto cross_a = [&]() {          // the request stream has used up its segment: on to the next one (or past the end)
        asm volatile("" ::: "memory");
        if (rseg == 0 && c1 > c0) {
            rseg = 1; rleft = c1 - c0; ra0 = arow(1, 0); ra1 = arow(1, 1);
            rb0 += koff[1] - koff[0] - c0 * S2_BK; rb1 += koff[1] - koff[0] - c0 * S2_BK;
        } else if (rseg <= 1 && nt > c1) {
            const int kprev = rseg == 0 ? koff[0] + c0 * S2_BK : koff[1] + (c1 - c0) * S2_BK;
            rseg = 2; rleft = nt - c1; ra0 = arow(2, 0); ra1 = arow(2, 1);
            rb0 += koff[2] - kprev; rb1 += koff[2] - kprev;
        } else rleft = 1 << 30;
    };
    auto advance = [&]() {
        ra0 += S2_BK * 4; ra1 += S2_BK * 4; rb0 += S2_BK; rb1 += S2_BK;
        if (--rleft == 0) cross_a();
    };
    // (the image pointer runs on its own: the planes of tile t + 2 are transferred a tile later than A's tile t + 3 is requested)
    int bseg = rseg, bleft = rleft;
    auto cross_b = [&]() {
        if (bseg == 0 && c1 > c0) {
            bseg = 1; bleft = c1 - c0; rbi += (long long)((koff[1] - koff[0]) / S2_BK - c0) * S2_BIMG_TILE;
        } else if (bseg <= 1 && nt > c1) {
            const int kprev = bseg == 0 ? koff[0] + c0 * S2_BK : koff[1] + (c1 - c0) * S2_BK;
            bseg = 2; bleft = nt - c1; rbi += (long long)((koff[2] - kprev) / S2_BK) * S2_BIMG_TILE;
        } else bleft = 1 << 30;
    };
    auto advance_b = [&]() {
        rbi += S2_BIMG_TILE;
        if (--bleft == 0) cross_b();
    };
    // Operand rows on their way in: LDS-DMA into the area the previous cell state will take at the end of the K loop (idle until
    // then) -- every lane its 16 bytes, piece `slot` of this wave's 1 KB: rows r0 / r0 + 128 of the stage's two K tiles of A
    // (slots 0..3) and, without a weight image, of B (slots 4..7).  The lane reads back what its own wave transferred (no barrier,
    // the wave's own vmcnt wait), splits it and stores the planes.  No operand value waits in a register while its stage's
    // products run, and there is no load whose destination registers the compiler does not know about.
    auto raw_in = [&](const char* src, int slot) {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)s2_smem)
                             + (unsigned)(S2_CB + slot * 8192) + (unsigned)__builtin_amdgcn_readfirstlane(wave * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    const char* const raw_lane = s2_smem + S2_CB + wave * 1024 + lane * 16;
    auto raw = [&](int slot) { return *reinterpret_cast<const f32x4*>(raw_lane + slot * 8192); };
    auto request = [&](int i) {           // K tile i (0 / 1) of the stage being requested: where the running pointers stand
        raw_in(ra0, 2 * i); raw_in(ra1, 2 * i + 1);
        if (!BIMG) { raw_in(reinterpret_cast<const char*>(rb0), 4 + 2 * i); raw_in(reinterpret_cast<const char*>(rb1), 4 + 2 * i + 1); }
    };
#define CASV_S2_LANDED asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // BIMG: plane p of the image tile whose (lane's) address is `tile` -> B plane p of LDS buffer `buf`, this wave's 1-KB piece
    auto dma_at = [&](const char* tile, int buf, int p) {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)s2_smem)
                             + (unsigned)(S2_TB + buf * S2_BUF + (3 + p) * S2_PLANE) + (unsigned)__builtin_amdgcn_readfirstlane(wave * 1024);
        unsigned keep;
        const unsigned long long t64 = (unsigned long long)(tile + p * S2_PLANE);       // (said to be uniform: a scalar register pair)
        const unsigned long long tu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(t64 >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)t64);     // (the builtin returns int: no sign extension of the low half)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(img_lane), "s"(tu), "s"(dst) : "memory");
    };
    auto dma_tile = [&](int buf, int p) { dma_at(rbi, buf, p); };

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
    // Fragments of the 32-deep instruction: lane l holds row (l & 15) of its 16-row block and the stage's k 8 (l >> 4) .. + 7 --
    // 16-byte piece (l >> 4) & 1 of K tile l >> 5, i.e. of LDS buffer l >> 5 (a row's two pieces are swapped where bit 4 of the
    // row is set: odd blocks)
    const int fr_lane = (lane >> 5) * S2_BUF + (lane & 15) * 32;
    const int fr_e = fr_lane + ((lane >> 4) & 1) * 16, fr_o = fr_lane + (((lane >> 4) & 1) ^ 1) * 16;
    // (four lane addresses; plane and block are 16-bit instruction offsets behind them)
    const char* const fa_e = s2_smem + S2_TB + wm * 64 * 32 + fr_e; const char* const fa_o = s2_smem + S2_TB + wm * 64 * 32 + fr_o;
    auto frag_a = [&](int plane, int rb) { return *reinterpret_cast<const bf16x8*>(((rb & 1) ? fa_o : fa_e) + plane * S2_PLANE + rb * 16 * 32); };
    // (B's lane addresses are A's plus a wave-uniform distance, added where a fragment is read: two registers instead of four)
    const int fb_delta = __builtin_amdgcn_readfirstlane(3 * S2_PLANE + (wn * 128 - wm * 64) * 32);
    auto frag_b = [&](int plane, int c) {
        return *reinterpret_cast<const bf16x8*>(((c & 1) ? fa_o : fa_e) + fb_delta + plane * S2_PLANE + c * 16 * 32);
    };
    char* const st_b0 = s2_smem + S2_TB + st_off; char* const st_b1 = s2_smem + S2_TB + S2_BUF + st_off;
    // staging in two windows: split four k of row r0 + 128 i, store two of the planes now, hand the third back
    auto split_store = [&](const f32x4 x, int buf, int plane0, int i, const bool hold_last, u32x2& held) {
        u32x2 p0, p1, p2;
        split4(x, p0, p1, p2);
        char* base = (buf ? st_b1 : st_b0) + plane0 * S2_PLANE + i * 128 * 32;
        if (hold_last) { *reinterpret_cast<u32x2*>(base) = p0; *reinterpret_cast<u32x2*>(base + S2_PLANE) = p1; held = p2; }          // A: a0, a1 now, a2 later
        else { *reinterpret_cast<u32x2*>(base + S2_PLANE) = p1; *reinterpret_cast<u32x2*>(base + 2 * S2_PLANE) = p2; held = p0; }     // B: b1, b2 now, b0 later
    };
    auto store_plane = [&](const u32x2 v, int buf, int plane, int i) {
        *reinterpret_cast<u32x2*>((buf ? st_b1 : st_b0) + plane * S2_PLANE + i * 128 * 32) = v;
    };
    auto zero_rows = [&](int buf, int plane0, int nplanes) {          // a stage's missing second K tile: zeros in both operands
        for (int p = plane0; p < plane0 + nplanes; ++p) { store_plane(u32x2{0u, 0u}, buf, p, 0); store_plane(u32x2{0u, 0u}, buf, p, 1); }
    };


    // ---- accumulators: 16x16 blocks [row block of 16][column block of 16] of the wave's 64 x 128 ----
    f32x4 acc16[4][8];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc16[rb][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // fragment registers: ONE B plane (8 column blocks), two A planes (4 row blocks each): fX / fY take turns as "a0 of the stage,
    // kept from its first product to its last" and "a1, then a2, then a0 of the next stage"
    bf16x8 fb[8], fX[4], fY[4];
#define CASV_S2_PROD(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, ACC, 0, 0, 0);
#define CASV_S2_FENCE __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: stage 0 (tiles 0 and 1) into LDS, stage 1 requested; a0 and b2 of stage 0 in registers ----
    if (nt > 0) {
        request(0); if (BIMG) { dma_tile(0, 0); dma_tile(0, 1); dma_tile(0, 2); advance_b(); } advance();
        if (nt > 1) { request(1); if (BIMG) { dma_tile(1, 0); dma_tile(1, 1); dma_tile(1, 2); advance_b(); } advance(); }
        CASV_S2_LANDED
        store_op(raw(0), raw(1), 0, 0); if (!BIMG) store_op(raw(4), raw(5), 0, 3);
        if (nt > 1) { store_op(raw(2), raw(3), 1, 0); if (!BIMG) store_op(raw(6), raw(7), 1, 3); }
        else zero_rows(1, 0, 6);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the read-backs have returned before the next transfers overwrite them)
    if (nt > 2) { request(0); advance(); }
    if (nt > 3) { request(1); advance(); }
    __syncthreads();
    if (nt > 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) fb[c] = frag_b(2, c);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) fX[rb] = frag_a(0, rb);
    }

    // One stage = two K tiles (k 0..15 from buffer 0, k 16..31 from buffer 1) = six products of 32 instructions per wave:
    //   p1 a0.b2, p2 a0.b1 | barrier X | p3 a1.b1, p4 a1.b0, p5 a2.b0 | barrier Y | p6 a0.b0
    // Q holds a0 of the stage; P takes a1 (read under p1), a2 (rolled in under p4), a0 of the next stage (read under p6);
    // fb rolls b2 -> b1 (under p1) -> b0 (under p3) -> b2 of the next stage (under p6); p2 and p5, in front of the barriers, read
    // nothing.  A plane's place in LDS is refilled IN PLACE with the same plane of the next stage once every wave has read it: behind
    // X the planes b2, a0, b1, a1 (window 1: the transfers of b2', b1', the split of A with the stores of a0', a1', the requests
    // for the stage after next), behind Y the planes b0, a2 (window 2: the stores of a2' -- held in registers since the split --
    // and the transfers of b0').  Y publishes window 1 (read from p6 on), the next X window 2 (read from the next p3 on).
#define CASV_S2_P_COUT(QA, ROLL_STMT)                                                                         \
    _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) {                                                        \
        _Pragma("unroll") for (int rb_ = 0; rb_ < 4; ++rb_) CASV_S2_PROD(QA[rb_], fb[c_], acc16[rb_][c_])     \
        ROLL_STMT                                                                                             \
        CASV_S2_FENCE                                                                                         \
    }
#define CASV_S2_P_ROUT(PA, ROLL_STMT)                                                                         \
    _Pragma("unroll") for (int rb_ = 0; rb_ < 4; ++rb_) {                                                     \
        _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) CASV_S2_PROD(PA[rb_], fb[c_], acc16[rb_][c_])        \
        ROLL_STMT                                                                                             \
        CASV_S2_FENCE                                                                                         \
    }
    // FULL: steady state (the two stages behind this one exist whole): no conditions, counted waits.
#define CASV_S2_STAGE(S, P, Q, FULL)                                                                          \
    {                                                                                                         \
        const bool next_ = FULL || 2 * (S) + 2 < nt, next1_ = FULL || 2 * (S) + 3 < nt;                       \
        const bool req0_ = FULL || 2 * (S) + 4 < nt, req1_ = FULL || 2 * (S) + 5 < nt;                        \
        CASV_S2_P_COUT(Q, { fb[c_] = frag_b(1, c_); if (c_ < 4) P[c_] = frag_a(1, c_); })           /* p1 */ \
        CASV_S2_P_COUT(Q, {})                                                                       /* p2 */ \
        CASV_S2_FENCE                                                                                         \
        CASV_S2_LANDED                       /* the requests of the stage before; window 2's transfers of this wave with them */ \
        __syncthreads();                                                                            /* X */  \
        if (next_) {                                                                         /* window 1 */  \
            if (BIMG) { tb0_ = rbi; dma_at(tb0_, 0, 2); dma_at(tb0_, 0, 1); advance_b();                      \
                        if (next1_) { tb1_ = rbi; dma_at(tb1_, 1, 2); dma_at(tb1_, 1, 1); advance_b(); } }    \
            split_store(raw(0), 0, 0, 0, true, h2_[0]); split_store(raw(1), 0, 0, 1, true, h2_[1]);               \
            if (!BIMG) { split_store(raw(4), 0, 3, 0, false, hb_[0]); split_store(raw(5), 0, 3, 1, false, hb_[1]); }  \
            if (next1_) {                                                                                     \
                split_store(raw(2), 1, 0, 0, true, h2_[2]); split_store(raw(3), 1, 0, 1, true, h2_[3]);       \
                if (!BIMG) { split_store(raw(6), 1, 3, 0, false, hb_[2]); split_store(raw(7), 1, 3, 1, false, hb_[3]); }  \
            } else { zero_rows(1, 0, 2); zero_rows(1, 4, 2); }                                                \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
            if (req0_) { request(0); advance(); }                                                             \
            if (req1_) { request(1); advance(); }                                                             \
        }                                                                                                     \
        CASV_S2_P_COUT(P, { fb[c_] = frag_b(0, c_); })                                              /* p3 */ \
        CASV_S2_P_ROUT(P, { P[rb_] = frag_a(2, rb_); })                                             /* p4 */ \
        CASV_S2_P_ROUT(P, {})                                                                       /* p5 */ \
        CASV_S2_FENCE                                                                                         \
        if (FULL) { if (BIMG) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }     /* this wave's four transfers of window 1; the four requests behind them stay in flight */ \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __syncthreads();                                                                            /* Y */  \
        if (!FULL && !req0_ && cell_pending) { cell_request(); cell_pending = false; }   /* (every wave has read its last operand rows back: the area is the cell state's from here) */ \
        if (next_) {                                                                         /* window 2 */  \
            store_plane(h2_[0], 0, 2, 0); store_plane(h2_[1], 0, 2, 1);                                       \
            if (!BIMG) { store_plane(hb_[0], 0, 3, 0); store_plane(hb_[1], 0, 3, 1); }                        \
            if (BIMG) dma_at(tb0_, 0, 0);                                                                     \
            if (next1_) {                                                                                     \
                store_plane(h2_[2], 1, 2, 0); store_plane(h2_[3], 1, 2, 1);                                   \
                if (!BIMG) { store_plane(hb_[2], 1, 3, 0); store_plane(hb_[3], 1, 3, 1); }                    \
                if (BIMG) dma_at(tb1_, 1, 0);                                                                 \
            } else zero_rows(1, 2, 2);                                                                        \
        }                                                                                                     \
        CASV_S2_P_COUT(Q, { if (next_) { fb[c_] = frag_b(2, c_); if (c_ < 4) P[c_] = frag_a(0, c_); } })   /* p6 */ \
    }
    // ---- previous cell state: every wave fetches the 64 rows x 32 units it will need itself, as LDS-DMA under the last stages (into
    // the area the operand rows were staged through, once the last of them has been read back) ----
    const bool plain = EPI == EPI_PLAIN || g.epi_plain;
    const bool cfirst = !plain && sgc.first_base && step == 0;
    const bool czero = !plain && sgc.skip_first && step == 0 && !cfirst;
    char* const cellw = s2_smem + S2_CB + wave * (64 * 128);
    bool cell_pending = !plain && !czero;
    auto cell_request = [&]() {
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
    };
    // The steady state written out group by group (four products of one column block, or half a row block's eight), a full scheduling
    // fence behind each: the group, at most one fragment read, and ONE piece of the windows' work -- a transfer, a read-back, half of a
    // value pair's split (5-6 vector instructions), the stores of a row once its four values are split -- so that the matrix pipe is
    // fed by this wave alone while its SIMD partner does the same.  As one run behind the barrier the ~100 vector instructions of a
    // window would leave the pipe idle in both waves at once (they pass the barrier together).
    f32x4 xa_[4], xb_[4]; unsigned qa_ = 0, qb_ = 0; f32x2 ra_ = {0.f, 0.f}, rb__ = {0.f, 0.f};
    u32x2 wa0_ = {0u, 0u}, wa1_ = {0u, 0u}, wa2_ = {0u, 0u}, wb0_ = {0u, 0u}, wb1_ = {0u, 0u}, wb2_ = {0u, 0u};
    u32x2 h2_[4] = {u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}}, hb_[4] = {u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}};
    const char* tb0_ = rbi; const char* tb1_ = rbi;
    auto st_read = [&](int v) { xa_[v] = raw(v); if (!BIMG) xb_[v] = raw(4 + v); };
    auto st_b = [&](int v, int h) { split_l1(xa_[v], h, qa_, ra_); if (!BIMG) split_l1(xb_[v], h, qb_, rb__); };
    auto st_c = [&](int v, int h) { split_l23(qa_, ra_, h, wa0_, wa1_, wa2_); if (!BIMG) split_l23(qb_, rb__, h, wb0_, wb1_, wb2_); };
    auto st_store = [&](int v) {
        store_plane(wa0_, v >> 1, 0, v & 1); store_plane(wa1_, v >> 1, 1, v & 1); h2_[v] = wa2_;
        if (!BIMG) { store_plane(wb1_, v >> 1, 4, v & 1); store_plane(wb2_, v >> 1, 5, v & 1); hb_[v] = wb0_; }
    };
    // window 1, slot n of 24 (behind the groups of p3, p4, p5)
    // (kind 0: a stage in the middle; 1: the last but one -- nothing is requested any more, the cell state is; 2: the last -- no windows)
    auto w1 = [&](int n, const bool steady, const int kind) {
        if (kind == 2 || (kind == 1 && n >= 18)) return;
        switch (n) {
        case 0: if (BIMG) dma_at(tb0_, 0, 2); st_read(0); break;
        case 1: if (BIMG) dma_at(tb1_, 1, 2); break;
        case 2: if (BIMG) dma_at(tb0_, 0, 1); st_b(0, 0); break;
        case 3: if (BIMG) dma_at(tb1_, 1, 1); st_c(0, 0); break;
        case 4: st_read(1); st_b(0, 1); break;
        case 5: st_c(0, 1); st_store(0); break;
        case 6: st_b(1, 0); break;
        case 7: st_c(1, 0); st_read(2); break;
        case 8: st_b(1, 1); break;
        case 9: st_c(1, 1); st_store(1); break;
        case 10: st_b(2, 0); st_read(3); break;
        case 11: st_c(2, 0); break;
        case 12: st_b(2, 1); break;
        case 13: st_c(2, 1); st_store(2); break;
        case 14: st_b(3, 0); break;
        case 15: st_c(3, 0); break;
        case 16: st_b(3, 1); break;
        case 17: st_c(3, 1); st_store(3); break;
        case 18: raw_in(ra0, 0); if (!BIMG) raw_in(reinterpret_cast<const char*>(rb0), 4); break;     // (the read-backs have long returned: their values are split)
        case 19: raw_in(ra1, 1); if (!BIMG) raw_in(reinterpret_cast<const char*>(rb1), 5);
                 if (steady) { ra0 += S2_BK * 4; ra1 += S2_BK * 4; rb0 += S2_BK; rb1 += S2_BK; } else advance(); break;
        case 20: raw_in(ra0, 2); if (!BIMG) raw_in(reinterpret_cast<const char*>(rb0), 6); break;
        case 21: raw_in(ra1, 3); if (!BIMG) raw_in(reinterpret_cast<const char*>(rb1), 7);
                 if (steady) { ra0 += S2_BK * 4; ra1 += S2_BK * 4; rb0 += S2_BK; rb1 += S2_BK; } else advance(); break;
        default: break;
        }
    };
    auto w2 = [&](int n, const int kind) {          // window 2, behind the groups of p6
        if (kind == 2) return;
        if (kind == 1 && n == 4 && cell_pending) { cell_request(); cell_pending = false; }      // (every wave has read its last operand rows back)
        switch (n) {
        case 0: store_plane(h2_[0], 0, 2, 0); store_plane(h2_[1], 0, 2, 1); if (BIMG) dma_at(tb0_, 0, 0); break;
        case 1: store_plane(h2_[2], 1, 2, 0); store_plane(h2_[3], 1, 2, 1); if (BIMG) dma_at(tb1_, 1, 0); break;
        case 2: if (!BIMG) { store_plane(hb_[0], 0, 3, 0); store_plane(hb_[1], 0, 3, 1); } break;
        case 3: if (!BIMG) { store_plane(hb_[2], 1, 3, 0); store_plane(hb_[3], 1, 3, 1); } break;
        default: break;
        }
    };
#define CASV_S2_GC(QA, C) _Pragma("unroll") for (int rb_ = 0; rb_ < 4; ++rb_) CASV_S2_PROD(QA[rb_], fb[C], acc16[rb_][C])
#define CASV_S2_GR(PA, HG) _Pragma("unroll") for (int c_ = 4 * ((HG) & 1); c_ < 4 * ((HG) & 1) + 4; ++c_) CASV_S2_PROD(PA[(HG) >> 1], fb[c_], acc16[(HG) >> 1][c_])
#define CASV_S2_STAGE_FULL(P, Q, STEADY, KIND)                                                                            \
    {                                                                                                         \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                          /* p1 */ \
            CASV_S2_GC(Q, g_) fb[g_] = frag_b(1, g_); if (g_ < 4) P[g_] = frag_a(1, g_); CASV_S2_FENCE }      \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) { CASV_S2_GC(Q, g_) CASV_S2_FENCE }        /* p2 */ \
        CASV_S2_LANDED                                                                                        \
        __syncthreads();                                                                            /* X */  \
        if (KIND == 2) {} else if (STEADY) { tb0_ = rbi; tb1_ = rbi + S2_BIMG_TILE; rbi += 2 * S2_BIMG_TILE; } \
        else { tb0_ = rbi; advance_b(); tb1_ = rbi; advance_b(); }                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                          /* p3 */ \
            CASV_S2_GC(P, g_) fb[g_] = frag_b(0, g_); w1(g_, STEADY, KIND); CASV_S2_FENCE }                                 \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                          /* p4 */ \
            CASV_S2_GR(P, g_) if (g_ & 1) P[g_ >> 1] = frag_a(2, g_ >> 1); w1(8 + g_, STEADY, KIND); CASV_S2_FENCE }        \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) { CASV_S2_GR(P, g_) w1(16 + g_, STEADY, KIND); CASV_S2_FENCE }   /* p5 */ \
        if (KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
        else if (KIND == 0 && BIMG) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      /* this wave's four plane transfers of window 1; the four requests behind them stay in flight */ \
        __syncthreads();                                                                            /* Y */  \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                          /* p6 */ \
            CASV_S2_GC(Q, g_) if (KIND != 2) { fb[g_] = frag_b(2, g_); if (g_ < 4) P[g_] = frag_a(0, g_); } w2(g_, KIND); CASV_S2_FENCE }   \
    }
    int s = 0;
#ifdef CASV_S2_CLOCK
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), cr0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Steady pairs of stages while the next four requests and the next four image tiles stay inside their K segments (a stage is
    // ONE basic block: the pointers just step; a stream that has used up its segment is re-based behind the pair); everything else --
    // a segment that ends inside a pair, the last three stages -- goes through the stage with conditions below, one at a time (it
    // leaves a0 of the next stage in fY: handed to fX).
    for (;;) {
        while (2 * s + 7 < nt && rleft >= 4 && (!BIMG || bleft >= 4)) {
            CASV_S2_STAGE_FULL(fY, fX, true, 0)
            CASV_S2_STAGE_FULL(fX, fY, true, 0)
            s += 2; rleft -= 4; bleft -= 4;
            if (rleft == 0) cross_a();
            if (BIMG && bleft == 0) cross_b();
        }
        if (2 * s + 4 == nt) {              // exactly two whole stages left: the same code without what reaches beyond the end
            CASV_S2_STAGE_FULL(fY, fX, false, 1)
            CASV_S2_STAGE_FULL(fX, fY, false, 2)
            s += 2;
        }
        if (2 * s >= nt) break;
        CASV_S2_STAGE(s, fY, fX, false)
        _Pragma("unroll") for (int rb = 0; rb < 4; ++rb) fX[rb] = fY[rb];
        s += 1;
    }
#ifdef CASV_S2_CLOCK
    const unsigned long long ph_t2 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        g_s2_clk[0] = __builtin_amdgcn_s_memtime() - ck0; g_s2_clk[1] = ph_t2 - cr0;
    }
#endif

    if (cell_pending) { __syncthreads(); cell_request(); }
#undef CASV_S2_STAGE
#undef CASV_S2_STAGE_FULL
#undef CASV_S2_GC
#undef CASV_S2_GR
#undef CASV_S2_P_COUT
#undef CASV_S2_P_ROUT
#undef CASV_S2_FENCE
#undef CASV_S2_PROD
#ifdef CASV_S2_CLOCK
    const unsigned long long ph_t3 = __builtin_amdgcn_s_memrealtime();
#endif
#undef CASV_S2_LANDED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the wave's own cell-state transfers (an LDS-DMA must not outlive its workgroup either)

    // ---- epilogue: lane l holds column (l & 15) of a 16-column block and the rows 4 (l >> 4) .. + 3 of a 16-row block ----
    const int l15 = lane & 15, lq = lane >> 4;
    if (plain) {
        float* cbase = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        float bcol[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bcol[c] = g.bias ? g.bias[n0 + wn * 128 + c * 16 + l15] : 0.0f;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            float* cb = cbase + (long long)(m0 + wm * 64 + rb * 16 + 4 * lq) * g.out.ld + n0 + wn * 128 + l15;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) cb[(long long)j * g.out.ld + c * 16] = acc16[rb][c][j] + bcol[c];
        }
        return;
    }
    const int nb = n0 + wn * 128;                 // this wave's 128 columns: gates i, f, c~, o of 32 units (column block 2 gate + unit half)
    float* cout = g.c_out.base + (long long)(step * g.c_out.step_mul + g.c_out.step_add) * g.c_out.slot_stride;
    float* hout = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
#pragma unroll
    for (int uh = 0; uh < 2; ++uh) {
        const int uw = uh * 16 + l15;             // hidden unit of this lane inside the wave's 32
        float bi = 0.f, bf_ = 0.f, bg = 0.f, bo = 0.f;
        if (g.bias) { bi = g.bias[nb + uw]; bf_ = g.bias[nb + 32 + uw]; bg = g.bias[nb + 64 + uw]; bo = g.bias[nb + 96 + uw]; }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            float* cb = cout + (long long)(m0 + wm * 64 + rb * 16 + 4 * lq) * g.c_out.ld + nb / 4 + uw;
            float* hb = hout + (long long)(m0 + wm * 64 + rb * 16 + 4 * lq) * g.out.ld + nb / 4 + uw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = rb * 16 + 4 * lq + j;
                const float cprev = czero ? 0.0f : *reinterpret_cast<const float*>(cellw + row * 128 + uw * 4);
                const LstmCellOut cell = lstm_cell(acc16[rb][uh][j] + bi, acc16[rb][2 + uh][j] + bf_, acc16[rb][4 + uh][j] + bg, acc16[rb][6 + uh][j] + bo, cprev);
                cb[(long long)j * g.c_out.ld] = cell.c;
                hb[(long long)j * g.out.ld] = cell.h;
            }
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
