// 256 x 256 persistent weight-gradient kernel for the wide bf16 convolutions (gfx950): the TN sibling of gemm256.hip.
//
//   dW[tap][n1][n2] = sum_m dY[m][n1] * X[m + tap - pad][n2]          (fp32 out; a tap that leaves the sample's window adds zero)
//
// Same contraction as gemm_tn_w2 (gemm.hip); reference call sites: autograd of modules/encoder.py:34 (95008 -> 1024),
// modules/decoder.py:117 (recon head 1024 -> 95008), modules/common.py:135-141 (decoder residual block k5 convs).
//
// Why a second kernel: gemm_tn_w2's 128 x 256 tile issues 1.5 x the LDS-DMA pieces per MFMA of a 256 x 256 tile and ran at
// 990-1090 TFLOP/s where gemm_nt_t256 reaches 1330 on the same FLOPs.  This kernel keeps gemm_nt_t256's pipeline unchanged --
// one 512-thread workgroup per CU, 8 waves = (row half g, 64-column block wq), 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 per wave,
// two 64 KiB K-tile buffers refilled in QUARTERS two sections after the section that read them, counted vmcnt + one barrier
// per section, the two row halves half a section out of phase, a persistent walk over an XCD-chunked item list with the DMA
// stream running ahead across item boundaries -- and changes what the TN contraction needs:
//  * the reduction index is the ROW index m of both operands, so a K-tile is 64 rows x 512 B of dY and of X exactly as they
//    lie in memory.  LDS image of a K-tile: 4 slabs per operand, slab s = columns [64 s, 64 s + 64) of the tile as 64 k-rows
//    x 128 B.  A DMA piece (1 KiB) is 8 k-rows of one slab = 8 whole cache lines (a first version with 32-column slabs moved
//    half lines: twice the L2 requests per byte); a quarter is two slabs = 16 pieces, two per wave.  The quarters are whole
//    slabs because the waves' column tiles are dealt accordingly: wave (g, wq) owns dY columns g*128 + [0, 128) (slab 2g in the
//    first half of a K-tile's sections, slab 2g+1 in the second) and X columns wq*32 + [0, 32) and 128 + wq*32 + [0, 32) (the
//    first pair of tiles from slabs 0-1, the second from slabs 2-3).
//  * MFMA operands come from ds_read_b64_tr_b16 (a 16-lane group reads 4 k-rows x 32 B and returns each lane its column's 4
//    values): two reads per operand.  Row r stores its four 32-byte pairs XOR-permuted by (bit 1 of r) | (bit 3 of r) << 1 (the
//    DMA applies it on the source side), so the 8 rows a half-instruction touches (r0..r0+3 and r0+8..r0+11) cover all 64 banks.
//  * taps are work items (each tap has its own output), the tap shift is a row shift of X folded into the DMA's scalar offset,
//    and an X row whose tap leaves its sample's window is pushed out of the buffer range per lane (a lane owns one k-row of a
//    piece), so it lands as zeros: no patch pass over LDS.
//  * the first MFMA operand is the X fragment: a lane ends with four consecutive n2 of one row n1 = 16 bytes of fp32 output.
// Item order: (slice, tile, tap) with tap fastest -- the taps of a tile read the same two panels -- and the column-tile index
// fastest among tiles, XCD-chunked.  Deterministic: split-K slices (over m) go to slabs summed in slice order by the caller.
#include <math.h>
#include <stdlib.h>
#include "sgv_common.h"

typedef __attribute__((address_space(3))) void q256_lds_t;
typedef float q256_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t q256_u4 __attribute__((ext_vector_type(4)));
typedef long q256_l2 __attribute__((ext_vector_type(2)));
constexpr uint32_t Q256_OOB = 0x7FFFFFF0u;

// MT: more than one tap (per-lane tap windows are tested when an X piece is issued)
// Work-stealing step of gemm_tn_t256_kernel<.., STEAL> (one thread): the next item of a list whose owner has not started, or -1.
// sm[1] = victims looked at so far (-1 on the first call), sm[0] = result.
__device__ __attribute__((noinline)) void q256_steal_next(int* sched, int* sm, int G, int self, int q8, int r8, bool had_list) {
    int item = -1;
    int* S = sched; int* Nc = sched + 256; int* C = sched + 512;
    int scan = sm[1];
    if (scan < 0) {
        // a workgroup without a list of its own gets here microseconds after the launch: give the others time to announce
        // themselves before taking anybody's list (12 x 64 x 64 cycles: about 20 us)
        if (!had_list) for (int d_ = 0; d_ < 12; ++d_) __builtin_amdgcn_s_sleep(64);
        scan = (__hip_atomic_load(C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= G) ? G : 0;     // everybody started: nothing to take
    }
    while (scan < G) {
        const int v = (int)(((unsigned)self + (unsigned)scan + 1u) % (unsigned)G);     // the last one looked at is the workgroup itself
        int st = __hip_atomic_load(&S[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st == 0) st = atomicCAS(&S[v], 0, 2) == 1 ? 1 : 2;
        if (st == 1) { ++scan; continue; }             // its owner works on it
        const int xv = v & 7, jv = v >> 3;
        const int lo_v = xv < r8 ? xv * (q8 + 1) : r8 * (q8 + 1) + (xv - r8) * q8;
        const int hi_v = lo_v + (xv < r8 ? q8 + 1 : q8);
        const int nb_v = (G - xv + 7) >> 3;
        const int k = atomicAdd(&Nc[v], 1);
        const long it = (long)lo_v + jv + (long)k * nb_v;
        if (it < hi_v) { item = (int)it; break; }      // stay on this victim
        ++scan;
    }
    sm[1] = scan;
    sm[0] = item;
}

// STEAL (GemmTN::sched != null; launched while another stream's resident workgroups -- a collective's channels -- may hold CUs):
// the static lists stay, but a workgroup that is not placed until others have finished does not hold the launch back.  Every
// workgroup announces itself at entry (sched[w]: 0 -> 1, one atomic before the pipeline starts); a workgroup that has finished
// its own list looks for lists whose owner has not started (0 -> 2: from then on that list is handed out item by item through
// the counter sched[256 + w], to thieves and to its late owner alike) and works them off one item at a time.  When every
// workgroup has announced itself by the time the first one finishes (sched[512] == gridDim.x) nothing is scanned.  No waiting
// on flags anywhere: every workgroup reaches its exit.
template <bool MT, bool STEAL = false>
__global__ __launch_bounds__(512) void gemm_tn_t256_kernel(const GemmTN p) {
    constexpr int NST = 32;                            // buffer stores per wave and epilogue
    __shared__ __attribute__((aligned(1024))) unsigned char smem[131072];
    __shared__ int sched_sm[2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wq = wave & 3;

    // ---------------- static schedule: XCD-chunked list of work items ----------------
    const int tiles_1 = (p.N1 + 255) >> 8, tiles_2 = (p.N2 + 255) >> 8;
    const int ntile = tiles_1 * tiles_2;
    const int kts = (p.M + 63) >> 6;
    const int per_z = ntile * p.taps;
    const int nitems = per_z * p.splitk;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int q8 = nitems >> 3, r8 = nitems & 7;
    const int it_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int it_hi = it_lo + (xcd < r8 ? q8 + 1 : q8);
    const int nbx = ((int)gridDim.x - xcd + 7) >> 3;
    // the list being worked on: items ls_first, ls_first + ls_stride, ... < ls_end (the static list first; STEAL: then single items)
    int ls_first = __builtin_amdgcn_readfirstlane(it_lo + jb), ls_stride = __builtin_amdgcn_readfirstlane(nbx), ls_end = __builtin_amdgcn_readfirstlane(it_hi);
    bool have = ls_first < ls_end;
    if constexpr (!STEAL) {
        if (!have) return;                                    // workgroup-uniform
    } else {
        if (tid == 0) {
            sched_sm[0] = atomicCAS(&p.sched[blockIdx.x], 0, 1);
            atomicAdd(&p.sched[512], 1);
            sched_sm[1] = -1;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(sched_sm[0]) != 0) have = false;      // placed late: thieves own the list (it is handed out through its counter)
        __syncthreads();
    }

    const int lda_b = (int)(p.lda * 2), ldx_b = (int)(p.ldb * 2);
    // the X descriptor starts `pad` rows BEFORE the buffer: the tap shift (tap - pad + pad >= 0 rows) goes to the scalar offset and
    // stays non-negative.  Lanes whose row leaves its sample's window get an offset beyond the extent (hardware zero-fill), and
    // every row in front of the buffer is such a row, so nothing in front of the buffer is ever read.
    const long x_shift = (long)p.pad * ldx_b;
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.B)) - x_shift, 0, (int)(p.b_bytes + x_shift), 0x00020000);

    // ---------------- DMA roles ----------------
    // a quarter = 2 slabs x 64 k-rows x 128 B = 16 pieces of 1 KiB (8 k-rows of a slab); wave w moves pieces 2w, 2w+1: slab
    // (w >> 2) of the quarter, k-rows (w & 3) * 16 + {0, 8} + (lane >> 3); LDS slot lane & 7 (16 B) of row r holds the source
    // chunk of pair (slot >> 1) ^ f(r), f(r) = (bit 1 of r) | (bit 3 of r) << 1: bit 3 is clear in the first piece, set in the second.
    const int dq = wave >> 2, kb = (wave & 3) * 16;
    const int drow = lane >> 3, dslot = lane & 7;
    const int df = (drow >> 1) & 1;
    const int dcol0 = ((((dslot >> 1) ^ df) << 4) + ((dslot & 1) << 3));                 // first piece: column of the chunk inside its slab
    const int dcol1 = ((((dslot >> 1) ^ (df | 2)) << 4) + ((dslot & 1) << 3));           // second piece (rows + 8)
    const int slabA0 = dq * 2, slabA1 = dq * 2 + 1;                                       // dY: columns g*128 + {0..63} / {64..127}, g = dq
    const int slabB0 = dq, slabB1 = 2 + dq;                                               // X: columns {0..127} / {128..255}
    unsigned char* const ldsA0 = smem + slabA0 * 8192 + kb * 128;
    unsigned char* const ldsA1 = smem + slabA1 * 8192 + kb * 128;
    unsigned char* const ldsB0 = smem + 32768 + slabB0 * 8192 + kb * 128;
    unsigned char* const ldsB1 = smem + 32768 + slabB1 * 8192 + kb * 128;
    const uint32_t dA8 = (uint32_t)(8 * lda_b), dX8 = (uint32_t)(8 * ldx_b);

    // per-item per-lane DMA state
    uint32_t vA0, vA0b, vA1, vA1b, vB0, vB0b, vB1, vB1b;     // (kb + drow [+ 8]) * ld + column bytes of the quarters' two pieces, or out of range
    int t_lane = 0;                        // MT: time index of row l_kt * 64 + kb + drow inside its sample
    // load cursor (wave-uniform)
    int li = ls_first;
    int l_kt = 0, l_kt_end = 0;
    int l_dt = 0;                          // tap - pad of item li
    uint32_t sP = 0, sQ = 0;               // scalar offsets: first row of the K-tile (X: + tap shift + pad) * ld
    bool l_active = true;

#define Q256_UNI(X) __builtin_amdgcn_readfirstlane(X)
    // item -> (slice z, tap, row tile t1 of n1, column tile t2 of n2): tap fastest, then t2 (order 2, the default) or t1
#define Q256_ITEM_OF(IT, Z, TAP, T1, T2)                                                                      \
    {                                                                                                         \
        Z = Q256_UNI((IT) / per_z);                                                                           \
        const int rem_ = (IT) - Z * per_z;                                                                    \
        const int tile_ = Q256_UNI(rem_ / p.taps);                                                            \
        TAP = rem_ - tile_ * p.taps;                                                                          \
        if (p.order == 2) { T1 = Q256_UNI(tile_ / tiles_2); T2 = tile_ - T1 * tiles_2; }                      \
        else { T2 = Q256_UNI(tile_ / tiles_1); T1 = tile_ - T2 * tiles_1; }                                   \
    }
#define Q256_SETUP_ITEM()                                                                                     \
    {                                                                                                         \
        int z_, tap_, t1_, t2_;                                                                               \
        Q256_ITEM_OF(li, z_, tap_, t1_, t2_)                                                                  \
        const int i0_ = t1_ << 8, j0_ = t2_ << 8;                                                             \
        l_kt = Q256_UNI((int)((long)kts * z_ / p.splitk));                                                    \
        l_kt_end = Q256_UNI((int)((long)kts * (z_ + 1) / p.splitk));                                          \
        l_dt = tap_ - p.pad;                                                                                  \
        sP = (uint32_t)(l_kt * 64) * (uint32_t)lda_b;                                                         \
        sQ = (uint32_t)(l_kt * 64 + tap_) * (uint32_t)ldx_b;                                                  \
        const uint32_t ra_ = (uint32_t)(kb + drow) * (uint32_t)lda_b, rx_ = (uint32_t)(kb + drow) * (uint32_t)ldx_b; \
        const int ca0_ = i0_ + slabA0 * 64, ca1_ = i0_ + slabA1 * 64, cb0_ = j0_ + slabB0 * 64, cb1_ = j0_ + slabB1 * 64; \
        vA0 = ca0_ + dcol0 < p.N1 ? ra_ + (uint32_t)((ca0_ + dcol0) * 2) : 0x80000000u;                       \
        vA0b = ca0_ + dcol1 < p.N1 ? ra_ + dA8 + (uint32_t)((ca0_ + dcol1) * 2) : 0x80000000u;                \
        vA1 = ca1_ + dcol0 < p.N1 ? ra_ + (uint32_t)((ca1_ + dcol0) * 2) : 0x80000000u;                       \
        vA1b = ca1_ + dcol1 < p.N1 ? ra_ + dA8 + (uint32_t)((ca1_ + dcol1) * 2) : 0x80000000u;                \
        vB0 = cb0_ + dcol0 < p.N2 ? rx_ + (uint32_t)((cb0_ + dcol0) * 2) : 0x80000000u;                       \
        vB0b = cb0_ + dcol1 < p.N2 ? rx_ + dX8 + (uint32_t)((cb0_ + dcol1) * 2) : 0x80000000u;                \
        vB1 = cb1_ + dcol0 < p.N2 ? rx_ + (uint32_t)((cb1_ + dcol0) * 2) : 0x80000000u;                       \
        vB1b = cb1_ + dcol1 < p.N2 ? rx_ + dX8 + (uint32_t)((cb1_ + dcol1) * 2) : 0x80000000u;                \
        if (MT) t_lane = (l_kt * 64 + kb + drow) % p.Tlen;                                                    \
    }
    // X offset of the lane's piece row (SECOND: the piece 8 rows further): pushed out of range when the tap leaves the row's sample
    // window (offsets + scalar offsets stay below 2^31, so an invalid lane stays >= 2^31 after the add)
#define Q256_VX(V, SECOND)                                                                                    \
    (MT ? (V) | ((unsigned)(((SECOND) ? (t_lane + 8 >= p.Tlen ? t_lane + 8 - p.Tlen : t_lane + 8) : t_lane) + l_dt) >= (unsigned)p.Tlen ? 0x80000000u : 0u) : (V))
#define Q256_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (q256_lds_t*)(DST), 16, (VOFF), (SOFF), 0, 0);
#define Q256_ISSUE_A0(PB, F) if ((F) || l_active) { Q256_DMA(rsP, vA0, sP, ldsA0 + (PB)) Q256_DMA(rsP, vA0b, sP, ldsA0 + (PB) + 1024) }
#define Q256_ISSUE_A1(PB, F) if ((F) || l_active) { Q256_DMA(rsP, vA1, sP, ldsA1 + (PB)) Q256_DMA(rsP, vA1b, sP, ldsA1 + (PB) + 1024) }
#define Q256_ISSUE_B0(PB, F) if ((F) || l_active) { Q256_DMA(rsQ, Q256_VX(vB0, 0), sQ, ldsB0 + (PB)) Q256_DMA(rsQ, Q256_VX(vB0b, 1), sQ, ldsB0 + (PB) + 1024) }
#define Q256_ISSUE_B1(PB, F) if ((F) || l_active) { Q256_DMA(rsQ, Q256_VX(vB1, 0), sQ, ldsB1 + (PB)) Q256_DMA(rsQ, Q256_VX(vB1b, 1), sQ, ldsB1 + (PB) + 1024) }
    // move the load cursor to the next K-tile of the stream
#define Q256_ADVANCE(F)                                                                                       \
    if ((F) || l_active) {                                                                                    \
        ++l_kt;                                                                                               \
        if ((F) || l_kt < l_kt_end) {                                                                         \
            sP += (uint32_t)(64 * lda_b); sQ += (uint32_t)(64 * ldx_b);                                       \
            if (MT) { t_lane += 64; if (t_lane >= p.Tlen) t_lane -= p.Tlen; }                                 \
        } else {                                                                                              \
            li += ls_stride;                                                                                  \
            if (li < ls_end) Q256_SETUP_ITEM() else l_active = false;                                         \
        }                                                                                                     \
    }

    // ---------------- fragment addressing ----------------
    // lane = 16 q + 4 qq + pp: k-group q (k = 8q .. 8q+7 of a 32-wide sub-step), k-row qq of the group's first / second four,
    // 8-byte segment pp of the tile's 32 bytes.  Tile pair P of a slab row sits at pair P ^ f, f = (bit 1 of qq) | (bit 0 of q) << 1.
    const int q = lane >> 4, lr = lane & 15;
    const int qq = lr >> 2, pp = lr & 3;
    const int ff = ((qq >> 1) & 1) | ((q & 1) << 1);
    const uint32_t smem_b = (uint32_t)(uintptr_t)(q256_lds_t*)smem;
    const uint32_t lrow = (uint32_t)((8 * q + qq) * 128 + 8 * pp);
    // dY row tile i of the wave's half: slab 2g + (i >> 2), pair i & 3
    const uint32_t fP0 = smem_b + g * 16384 + lrow + ((0 ^ ff) << 5), fP1 = smem_b + g * 16384 + lrow + ((1 ^ ff) << 5);
    const uint32_t fP2 = smem_b + g * 16384 + lrow + ((2 ^ ff) << 5), fP3 = smem_b + g * 16384 + lrow + ((3 ^ ff) << 5);
    // X column tile nt of the wave: slab (nt >> 1) * 2 + (wq >> 1), pair (wq & 1) * 2 + (nt & 1)
    const uint32_t fQ0 = smem_b + 32768 + (wq >> 1) * 8192 + lrow + ((((wq & 1) * 2 + 0) ^ ff) << 5);
    const uint32_t fQ1 = smem_b + 32768 + (wq >> 1) * 8192 + lrow + ((((wq & 1) * 2 + 1) ^ ff) << 5);
    const uint32_t fP0n = fP0 + 65536, fP1n = fP1 + 65536, fP2n = fP2 + 65536, fP3n = fP3 + 65536, fQ0n = fQ0 + 65536, fQ1n = fQ1 + 65536;

    q256_f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = (q256_f4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa00, fa01, fa10, fa11, fa20, fa21, fa30, fa31;     // dY fragments [row tile 0..3 of the half being read][k sub-step]
    bf16x8 fb00, fb01, fb10, fb11;                             // X fragments, columns 0-31: [col tile][k sub-step]
    bf16x8 fc00, fc01, fc10, fc11;                             // X fragments, columns 32-63

    // one operand = k-rows r..r+3 and r+4..r+7 of the lane's k-group: two transposed reads, 512 bytes (4 rows) apart
#define Q256_TR(DST, ADDR, IMM)                                                                               \
    {                                                                                                         \
        long lo_, hi_;                                                                                        \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"           \
                     : "=&v"(lo_), "=&v"(hi_) : "v"(ADDR), "n"(IMM), "n"((IMM) + 512));                       \
        q256_l2 pr_; pr_[0] = lo_; pr_[1] = hi_;                                                              \
        DST = __builtin_bit_cast(bf16x8, pr_);                                                                \
    }
    // dY row tiles R0..R0+3 of the wave's half (R0 = 0 or 4): slab 2g + (R0 >> 2), pairs 0..3; k sub-step = +4096
#define Q256_READ_A(P0, P1, P2, P3, R0)                                                                       \
    {                                                                                                         \
        Q256_TR(fa00, P0, ((R0) >> 2) * 8192 + 0) Q256_TR(fa01, P0, ((R0) >> 2) * 8192 + 4096)                \
        Q256_TR(fa10, P1, ((R0) >> 2) * 8192 + 0) Q256_TR(fa11, P1, ((R0) >> 2) * 8192 + 4096)                \
        Q256_TR(fa20, P2, ((R0) >> 2) * 8192 + 0) Q256_TR(fa21, P2, ((R0) >> 2) * 8192 + 4096)                \
        Q256_TR(fa30, P3, ((R0) >> 2) * 8192 + 0) Q256_TR(fa31, P3, ((R0) >> 2) * 8192 + 4096)                \
    }
    // X column tiles C0, C0+1 of the wave (C0 = 0 or 2): slab (C0 >> 1) * 2 + (wq >> 1)
#define Q256_READ_B(X, Q0, Q1, C0)                                                                            \
    {                                                                                                         \
        Q256_TR(X##00, Q0, ((C0) >> 1) * 16384 + 0) Q256_TR(X##01, Q0, ((C0) >> 1) * 16384 + 4096)            \
        Q256_TR(X##10, Q1, ((C0) >> 1) * 16384 + 0) Q256_TR(X##11, Q1, ((C0) >> 1) * 16384 + 4096)            \
    }
#define Q256_WAIT_A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa00), "+v"(fa01), "+v"(fa10), "+v"(fa11), "+v"(fa20), "+v"(fa21), "+v"(fa30), "+v"(fa31));
#define Q256_WAIT_B(X) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X##00), "+v"(X##01), "+v"(X##10), "+v"(X##11));
#define Q256_MMA(I, N, X, NI, FA, S) acc[I][N] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X##NI##S, FA##S, acc[I][N], 0, 0, 0);
    // 16 MFMAs: row tiles R0..R0+3 x column tiles C0, C0+1 (X fragments X) x 2 k sub-steps
#define Q256_MMA16(R0, C0, X)                                                                                 \
    {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        Q256_MMA(R0 + 0, C0 + 0, X, 0, fa0, 0) Q256_MMA(R0 + 0, C0 + 1, X, 1, fa0, 0)                         \
        Q256_MMA(R0 + 1, C0 + 0, X, 0, fa1, 0) Q256_MMA(R0 + 1, C0 + 1, X, 1, fa1, 0)                         \
        Q256_MMA(R0 + 2, C0 + 0, X, 0, fa2, 0) Q256_MMA(R0 + 2, C0 + 1, X, 1, fa2, 0)                         \
        Q256_MMA(R0 + 3, C0 + 0, X, 0, fa3, 0) Q256_MMA(R0 + 3, C0 + 1, X, 1, fa3, 0)                         \
        Q256_MMA(R0 + 0, C0 + 0, X, 0, fa0, 1) Q256_MMA(R0 + 0, C0 + 1, X, 1, fa0, 1)                         \
        Q256_MMA(R0 + 1, C0 + 0, X, 0, fa1, 1) Q256_MMA(R0 + 1, C0 + 1, X, 1, fa1, 1)                         \
        Q256_MMA(R0 + 2, C0 + 0, X, 0, fa2, 1) Q256_MMA(R0 + 2, C0 + 1, X, 1, fa2, 1)                         \
        Q256_MMA(R0 + 3, C0 + 0, X, 0, fa3, 1) Q256_MMA(R0 + 3, C0 + 1, X, 1, fa3, 1)                         \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // end of an L section: my DMAs older than the last four sections have landed; publish (see gemm256.hip)
#define Q256_LEND(F)                                                                                          \
    {                                                                                                         \
        if ((F) || wmode == 0) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");                  \
        else if (wmode == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(8 + NST) : "memory");     \
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // reads ordered by first use; every MFMA pair waits only for the fragments it consumes (LDS returns in order; lgkmcnt counts
    // at most 15, so the first wait of a section is looser than its operands need and still correct)
#ifndef Q256_FINEWAIT
#define Q256_FINEWAIT 1
#endif
#if Q256_FINEWAIT
#define Q256_W3(N, X, Y, Z) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X), "+v"(Y), "+v"(Z));
#define Q256_W2(N, X, Y) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X), "+v"(Y));
#define Q256_W1(N, X) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X));
#define Q256_MM2(R, C0, X, FA, S) Q256_MMA(R, C0 + 0, X, 0, FA, S) Q256_MMA(R, C0 + 1, X, 1, FA, S)
    // section 1: X tiles 0,1 + dY row tiles 0-3, k sub-step 0 first (24 reads)
#define Q256_SEC1_READ(P0, P1, P2, P3, Q0, Q1)                                                                \
        Q256_TR(fb00, Q0, 0) Q256_TR(fb10, Q1, 0) Q256_TR(fa00, P0, 0) Q256_TR(fa10, P1, 0)                   \
        Q256_TR(fa20, P2, 0) Q256_TR(fa30, P3, 0)                                                             \
        Q256_TR(fb01, Q0, 4096) Q256_TR(fb11, Q1, 4096) Q256_TR(fa01, P0, 4096) Q256_TR(fa11, P1, 4096)       \
        Q256_TR(fa21, P2, 4096) Q256_TR(fa31, P3, 4096)
#define Q256_SEC1_MMA()                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        Q256_W3(15, fb00, fb10, fa00) __builtin_amdgcn_sched_barrier(0); Q256_MM2(0, 0, fb, fa0, 0)           \
        Q256_W1(15, fa10) __builtin_amdgcn_sched_barrier(0); Q256_MM2(1, 0, fb, fa1, 0)                       \
        Q256_W1(14, fa20) __builtin_amdgcn_sched_barrier(0); Q256_MM2(2, 0, fb, fa2, 0)                       \
        Q256_W1(12, fa30) __builtin_amdgcn_sched_barrier(0); Q256_MM2(3, 0, fb, fa3, 0)                       \
        Q256_W3(6, fb01, fb11, fa01) __builtin_amdgcn_sched_barrier(0); Q256_MM2(0, 0, fb, fa0, 1)            \
        Q256_W1(4, fa11) __builtin_amdgcn_sched_barrier(0); Q256_MM2(1, 0, fb, fa1, 1)                        \
        Q256_W1(2, fa21) __builtin_amdgcn_sched_barrier(0); Q256_MM2(2, 0, fb, fa2, 1)                        \
        Q256_W1(0, fa31) __builtin_amdgcn_sched_barrier(0); Q256_MM2(3, 0, fb, fa3, 1)                        \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);
    // section 2: X tiles 2,3 (8 reads)
#define Q256_SEC2_READ(Q0, Q1)                                                                                \
        Q256_TR(fc00, Q0, 16384) Q256_TR(fc10, Q1, 16384) Q256_TR(fc01, Q0, 16384 + 4096) Q256_TR(fc11, Q1, 16384 + 4096)
#define Q256_SEC2_MMA()                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        Q256_W2(4, fc00, fc10) __builtin_amdgcn_sched_barrier(0);                                             \
        Q256_MM2(0, 2, fc, fa0, 0) Q256_MM2(1, 2, fc, fa1, 0) Q256_MM2(2, 2, fc, fa2, 0) Q256_MM2(3, 2, fc, fa3, 0) \
        Q256_W2(0, fc01, fc11) __builtin_amdgcn_sched_barrier(0);                                             \
        Q256_MM2(0, 2, fc, fa0, 1) Q256_MM2(1, 2, fc, fa1, 1) Q256_MM2(2, 2, fc, fa2, 1) Q256_MM2(3, 2, fc, fa3, 1) \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);
    // section 3: dY row tiles 4-7 (16 reads)
#define Q256_SEC3_READ(P0, P1, P2, P3)                                                                        \
        Q256_TR(fa00, P0, 8192) Q256_TR(fa10, P1, 8192) Q256_TR(fa20, P2, 8192) Q256_TR(fa30, P3, 8192)       \
        Q256_TR(fa01, P0, 8192 + 4096) Q256_TR(fa11, P1, 8192 + 4096) Q256_TR(fa21, P2, 8192 + 4096) Q256_TR(fa31, P3, 8192 + 4096)
#define Q256_SEC3_MMA()                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        Q256_W1(14, fa00) __builtin_amdgcn_sched_barrier(0); Q256_MM2(4, 2, fc, fa0, 0)                       \
        Q256_W1(12, fa10) __builtin_amdgcn_sched_barrier(0); Q256_MM2(5, 2, fc, fa1, 0)                       \
        Q256_W1(10, fa20) __builtin_amdgcn_sched_barrier(0); Q256_MM2(6, 2, fc, fa2, 0)                       \
        Q256_W1(8, fa30) __builtin_amdgcn_sched_barrier(0); Q256_MM2(7, 2, fc, fa3, 0)                        \
        Q256_W1(6, fa01) __builtin_amdgcn_sched_barrier(0); Q256_MM2(4, 2, fc, fa0, 1)                        \
        Q256_W1(4, fa11) __builtin_amdgcn_sched_barrier(0); Q256_MM2(5, 2, fc, fa1, 1)                        \
        Q256_W1(2, fa21) __builtin_amdgcn_sched_barrier(0); Q256_MM2(6, 2, fc, fa2, 1)                        \
        Q256_W1(0, fa31) __builtin_amdgcn_sched_barrier(0); Q256_MM2(7, 2, fc, fa3, 1)                        \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);
#else
#define Q256_SEC1_READ(P0, P1, P2, P3, Q0, Q1) Q256_READ_A(P0, P1, P2, P3, 0) Q256_READ_B(fb, Q0, Q1, 0)
#define Q256_SEC1_MMA() Q256_WAIT_A() Q256_WAIT_B(fb) __builtin_amdgcn_sched_barrier(0); Q256_MMA16(0, 0, fb)
#define Q256_SEC2_READ(Q0, Q1) Q256_READ_B(fc, Q0, Q1, 2)
#define Q256_SEC2_MMA() Q256_WAIT_B(fc) __builtin_amdgcn_sched_barrier(0); Q256_MMA16(0, 2, fc)
#define Q256_SEC3_READ(P0, P1, P2, P3) Q256_READ_A(P0, P1, P2, P3, 4)
#define Q256_SEC3_MMA() Q256_WAIT_A() __builtin_amdgcn_sched_barrier(0); Q256_MMA16(4, 2, fc)
#endif
    // one K-tile (section structure and refill distances of gemm256.hip): parity buffer PB is consumed; quarters of the
    // stream's next K-tiles go into PN (B1, A1: the K-tile the cursor points at) and, after the cursor has moved, into PB (A0, B0)
#define Q256_KTILE(P0, P1, P2, P3, Q0, Q1, PB, PN, F)                                                         \
    {                                                                                                         \
        Q256_SEC1_READ(P0, P1, P2, P3, Q0, Q1)                                                                \
        Q256_ISSUE_B1(PN, F)                                                                                  \
        if (early) Q256_LEND(F)                                                                               \
        Q256_SEC1_MMA()                                                                                       \
        if (!early) Q256_LEND(F)                                                                              \
        Q256_SEC2_READ(Q0, Q1)                                                                                \
        Q256_ISSUE_A1(PN, F)                                                                                  \
        if (early) Q256_LEND(F)                                                                               \
        Q256_SEC2_MMA()                                                                                       \
        if (!early) Q256_LEND(F)                                                                              \
        Q256_SEC3_READ(P0, P1, P2, P3)                                                                        \
        Q256_ADVANCE(F)                                                                                       \
        if (!(F) && !l_active) wmode = 2;                                                                     \
        Q256_ISSUE_A0(PB, F)                                                                                  \
        if (early) Q256_LEND(F)                                                                               \
        Q256_SEC3_MMA()                                                                                       \
        if (!early) Q256_LEND(F)                                                                              \
        Q256_ISSUE_B0(PB, F)                                                                                  \
        if (early) Q256_LEND(F)                                                                               \
        Q256_MMA16(4, 0, fb)                                                                                  \
        if (!early) Q256_LEND(F)                                                                              \
    }
    const bool early = g == 0;           // the two row halves run half a section out of phase (gemm256.hip)

    // ---------------- compute cursor ----------------
    int ci = ls_first;
    int c_z = 0, c_tap = 0, c_i0 = 0, c_j0 = 0, c_nkt = 0;
#define Q256_DECODE_C()                                                                                       \
    {                                                                                                         \
        int t1_, t2_;                                                                                         \
        Q256_ITEM_OF(ci, c_z, c_tap, t1_, t2_)                                                                \
        c_i0 = t1_ << 8; c_j0 = t2_ << 8;                                                                     \
        c_nkt = Q256_UNI((int)((long)kts * (c_z + 1) / p.splitk) - (int)((long)kts * c_z / p.splitk));        \
    }
    for (;;) {          // lists (one pass without STEAL)
    if (have) {
    li = ls_first; l_kt = 0; l_kt_end = 0; l_dt = 0; sP = 0; sQ = 0; l_active = true; ci = ls_first;
    Q256_DECODE_C()

    // ---------------- prologue: K-tile 0 entirely, the A0 / B0 quarters of K-tile 1 ----------------
    Q256_SETUP_ITEM()
    Q256_ISSUE_A0(0, 0) Q256_ISSUE_B0(0, 0) Q256_ISSUE_B1(0, 0) Q256_ISSUE_A1(0, 0)
    Q256_ADVANCE(0)
    Q256_ISSUE_A0(65536, 0) Q256_ISSUE_B0(65536, 0)
    if (l_active) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    int kt = 0;
    uint32_t pb = 0;                 // parity buffer of the K-tile being consumed
    int wmode = l_active ? 0 : 2;
    for (;;) {
        // fast path: two K-tiles with compile-time parity while neither cursor meets a boundary
        while (pb == 0 && wmode == 0 && kt + 2 < c_nkt && l_kt + 2 < l_kt_end) {
            Q256_KTILE(fP0, fP1, fP2, fP3, fQ0, fQ1, 0, 65536, 1)
            Q256_KTILE(fP0n, fP1n, fP2n, fP3n, fQ0n, fQ1n, 65536, 0, 1)
            kt += 2;
        }
        {
            const uint32_t pn = pb ^ 65536u;
            const uint32_t p0_ = fP0 + pb, p1_ = fP1 + pb, p2_ = fP2 + pb, p3_ = fP3 + pb, q0_ = fQ0 + pb, q1_ = fQ1 + pb;
            Q256_KTILE(p0_, p1_, p2_, p3_, q0_, q1_, pb, pn, 0)
            ++kt;
            pb = pn;
        }
        if (kt == c_nkt) {
            // ================= epilogue of item ci: raw fp32 sums, 16 bytes per lane and tile =================
            const int mw = c_i0 + g * 128, nw = c_j0 + wq * 32;          // column tiles 0,1 at nw, tiles 2,3 at nw + 128
            if (p.out_bf16) {      // same 32 stores per wave, 8 bytes per lane: round to nearest even, as the data-parallel wire copy does
                unsigned short* outp = reinterpret_cast<unsigned short*>(p.out) + (long)c_tap * p.out_tap_stride;
                const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)(((long)(p.N1 - 1) * p.ldo + p.N2) * 2), 0x00020000);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = mw + i * 16 + lr;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int col = nw + (nt >> 1) * 128 + (nt & 1) * 16 + q * 4;
                        const bool ok = row < p.N1 && col < p.N2;
                        const q256_f4 v = acc[i][nt];
                        acc[i][nt] = (q256_f4){0.f, 0.f, 0.f, 0.f};
                        typedef __bf16 q256_bf4 __attribute__((ext_vector_type(4)));
                        typedef uint32_t q256_u2 __attribute__((ext_vector_type(2)));
                        q256_bf4 o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(q256_u2, o), rsC, ok ? (uint32_t)(((long)row * p.ldo + col) * 2) : Q256_OOB, 0, 0);
                    }
                }
            } else {
            float* outp = p.out + (long)c_z * p.out_slab_stride + (long)c_tap * p.out_tap_stride;
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)(((long)(p.N1 - 1) * p.ldo + p.N2) * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = mw + i * 16 + lr;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int col = nw + (nt >> 1) * 128 + (nt & 1) * 16 + q * 4;
                    const bool ok = row < p.N1 && col < p.N2;
                    const q256_f4 v = acc[i][nt];
                    acc[i][nt] = (q256_f4){0.f, 0.f, 0.f, 0.f};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(q256_u4, v), rsC, ok ? (uint32_t)(((long)row * p.ldo + col) * 4) : Q256_OOB, 0, 0);
                }
            }
            }
            ci += ls_stride;
            const bool more = ci < ls_end;
            if (more) Q256_DECODE_C()
            if (wmode != 2) wmode = 1;
            __builtin_amdgcn_sched_barrier(0);
            if (!more) break;
            kt = 0;
        } else if (wmode == 1) {
            wmode = 0;               // the K-tile after an epilogue is over
        }
    }
    }   // have
    if constexpr (!STEAL) return;
    else {
        // ---- the next orphaned item, if any (thread 0 asks, everybody follows) ----
        __syncthreads();                                       // every wave is past its last LDS read
        if (tid == 0) q256_steal_next(p.sched, sched_sm, (int)gridDim.x, (int)blockIdx.x, q8, r8, it_lo + jb < it_hi);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(sched_sm[0]);      // wave-uniform: scalar registers
        if (item < 0) return;
        ls_first = item; ls_stride = nitems + 1; ls_end = item + 1; have = true;
        __syncthreads();
    }
    }   // lists
}

// =========================================================================================
// host side
// =========================================================================================
static inline int q256_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// bf16, both widths >= 256, at least 16 K-tiles of 64 rows, sample length >= 64 (one conditional subtract keeps a lane's time index)
bool gemm_tn256_eligible(int dtype, const GemmTN& p) {
    static const int on = getenv("SGV_GEMM_TN256") ? atoi(getenv("SGV_GEMM_TN256")) : 1;
    if (!on || dtype != 1 || p.cv_kw > 0) return false;
    // M % 64 == 0: a K-tile never runs past the last row (the row advance is the DMA's SCALAR offset, which the buffer range check
    // does not see; every other out-of-window row is pushed out of range through the per-lane offset)
    if (p.N1 < 256 || p.N2 < 256 || p.M < 1024 || p.M % 64 || p.Tlen < 64 || p.taps > 15) return false;
    if (p.N1 % 8 || p.N2 % 8 || p.lda % 8 || p.ldb % 8 || p.ldo % 4) return false;
    return true;
}
// the planner's choice for a weight gradient: the 256 x 256 kernel when its items fill the chip (>= 200 per slice) and the
// product is big enough to amortise the per-item epilogue (256 KiB of fp32 per 50 K-tiles at M = 3200)
bool gemm_tn_uses_t256(int dtype, const GemmTN& p) {
    static const double min_gf = getenv("SGV_TN256_MIN_GF") ? atof(getenv("SGV_TN256_MIN_GF")) : 150.0;
    if (!gemm_tn256_eligible(dtype, p)) return false;
    const long items = (long)q256_cdiv(p.N1, 256) * q256_cdiv(p.N2, 256) * p.taps;
    return items >= 200 && 2.0e-9 * p.M * p.N1 * p.N2 * p.taps >= min_gf;
}
int launch_gemm_tn256(const GemmTN& p, hipStream_t s) {
    if (!gemm_tn256_eligible(1, p)) return -1;
    if (p.splitk < 1 || (p.splitk > 1 && p.out_slab_stride <= 0)) return -1;
    if (p.out_bf16 && (p.splitk != 1 || p.ldo % 4)) return -1;         // bf16 output: direct stores only (slabs stay fp32)
    if (q256_cdiv(p.M, 64) / p.splitk < 4) return -1;                 // every item keeps >= 4 K-tiles (prologue + counted waits)
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) || ((uintptr_t)p.out & 15)) return -1;
    GemmTN q = p;
    static const int order_env = getenv("SGV_TN256_ORDER") ? atoi(getenv("SGV_TN256_ORDER")) : -1;
    q.order = order_env >= 0 ? order_env : 2;        // measured (one box, order 0 / 2): 665 / 650, 652 / 620, 837 / 826 us on the three big shapes
    q.a_bytes = ((long)(p.M - 1) * p.lda + p.N1) * 2;
    q.b_bytes = ((long)(p.M - 1) * p.ldb + p.N2) * 2;
    // 32-bit buffer offsets; a lane's offset + scalar offset (incl. the tap shift and 63 rows of run-ahead) must stay below 2^31
    if (q.a_bytes + 64L * p.lda * 2 >= 0x7FFFFFF0L || q.b_bytes + (64L + 2 * p.taps) * p.ldb * 2 >= 0x7FFFFFF0L) return -1;
    if (((long)(p.N1 - 1) * p.ldo + p.N2) * 4 >= 0x7FFFFFF0L) return -1;
    const int nitems = q256_cdiv(p.N1, 256) * q256_cdiv(p.N2, 256) * p.taps * p.splitk;
    int grid = ((nitems + 7) / 8) * 8;
    if (grid > 256) grid = 256;
    // work stealing pays from about three rounds of items on (2 rounds: 381 vs 372 us under occupancy, tests/micro/occupy_ab.sh)
    if (p.sched && (nitems >= 768 || p.force_w2 == 3)) {
        if (hipMemsetAsync(p.sched, 0, 513 * sizeof(int), s) != hipSuccess) return -2;
        if (p.taps > 1) hipLaunchKernelGGL((gemm_tn_t256_kernel<true, true>), dim3(grid), dim3(512), 0, s, q);
        else hipLaunchKernelGGL((gemm_tn_t256_kernel<false, true>), dim3(grid), dim3(512), 0, s, q);
    } else if (p.taps > 1) hipLaunchKernelGGL((gemm_tn_t256_kernel<true, false>), dim3(grid), dim3(512), 0, s, q);
    else hipLaunchKernelGGL((gemm_tn_t256_kernel<false, false>), dim3(grid), dim3(512), 0, s, q);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
