// Fused single-head self-attention core for the 16 x 16 maps (N = 256 tokens, C = 128 ... 1024 channels), bf16:
//     O[b][i][c] = sum_j softmax_j(C^-1/2 q[b][i] . k[b][j]) v[b][j][c]                                     (models/unet.py:176-189)
// in ONE kernel instead of Q.K^T (GEMM, fp32 S to HBM) -> softmax_rows (S -> P) -> P.V (GEMM): S and P never leave the CU.
//
// Workgroup = (image, block of 64 queries), 8 waves.  Operands come in the layouts the projection GEMMs already produce:
//   qk  [B][256][2C]   token-major, q in columns [0, C), k in [C, 2C)     (the fused q|k 1x1 conv)
//   vT  [B][C][256]    channel-major                                       (the v 1x1 conv, stored through Y_NCHW)
//   o   [B][256][C]    token-major (input of proj_out)
//
// Phase 1   S^T[key][query] = K . Q^T over the C channels in steps of 64: both operands are [row][channel] images (128-byte rows, 16-byte
//           unit u of row r in slot u ^ ((r >> 1) & 7), as conv_gemm_kernel.h), staged by LDS-DMA through a ring of three 40 KB stages
//           (256 key rows + 64 query rows), two stages in flight, counted vmcnt, one raw barrier per step.  Waves = 4 (key blocks of 64) x 2
//           (query blocks of 32).  Computing the TRANSPOSE puts, in every lane, four consecutive KEYS of one query: the softmax statistics of a
//           query reduce over registers, then over the four lane quarters (xor 16 / 32 shuffles), then over the four key-block waves through
//           256 floats of LDS; and P goes to LDS as 8-byte pieces of a [query][key] image -- the A/B-operand layout of phase 2.
// Softmax   max and sum per query in fp32 (exp2 of the log2(e)-scaled scores), P = bf16(e / sum).
// Phase 2   O^T[channel][query] = V^T . P^T over the 256 keys in steps of 32: V^T rows are channels ([row][key] image, 64-byte rows in the conv
//           kernel's rotated layout), P rows are queries (512-byte rows, unit slot ^ (row & 15)); ring of three 32 KB V^T chunks.  Waves = 8
//           channel blocks of C / 8.  The transpose again leaves four consecutive CHANNELS of a query in a lane: 8-byte global stores.
//           C > 512 (BASELINE configs[2]: the 768-channel AttnBlocks of the 128 x 128 model): phase 2 runs twice, over the lower and the upper half
//           of the channels, from the same P (a 32-key chunk of all 768 rows would be 48 KB, three of them do not fit beside P).
// The first V^T chunks are fetched while the softmax runs.  LDS: phase 1 3 x 40 KB; phase 2 P 32 KB + 3 x 32 KB; + 1 KB of statistics.
#pragma once
#include "conv_kernel.h"
#include "gn_arrive.h"

namespace wdm {

struct AttnFusedArgs {
    const void* q;       // [B][256][q_ld] token-major queries (bf16 / fp16)
    const void* k;       // [B][256][k_ld] token-major keys
    const void* v;       // v_tok = 0: V^T [B][C][256] channel-major; v_tok = 1: V [B][256][v_ld] token-major (read with the transposing LDS load)
    void* o;             // [B][256][C]
    const float* vbias;  // [C] or nullptr: added to the output rows (the rows of P sum to 1, so P.(V + 1 b^T) = P.V + 1 b^T: V^T is stored without it)
    int B, C;
    int q_ld, k_ld, v_ld;    // elements per token row
    float alpha;         // C^-1/2
    unsigned q_bytes, k_bytes, v_bytes;      // extents behind q / k / v (buffer descriptors)
    // QPROJ (round 5, folded AttnBlock only): the query projection q' = Mq . h + cq runs inside the kernel as phase 0 -- q is then unused, the workgroup's 64 query rows
    // are rows of k (the normalised input itself)
    const void* qw;      // [C][qw_ld] 16-bit: Mq = Wk^T Wq (rows = output channels)
    const float* qbias;  // [C]: cq = Wk^T bq
    int qw_ld;
    unsigned qw_bytes;
    int qw_slab;         // 0: plain [C][qw_ld]; else slab-major [C / 32][rows][32], elements between slabs
    // block-diagonal form (round 5; the 8 x 8 maps' AttnBlock, N = 64 tokens): an "image" of the kernel is a group of FOUR real images (4 x 64 = 256 token rows, contiguous in
    // [B][64][C] tensors), query block qb is real image 4 b + qb and attends to key block qb only -- the other three key blocks' scores are set to -inf before the softmax
    // (their P is exactly 0, so O and everything behind it are the real image's attention); B = groups, nimg = real images (a last group may be ragged)
    int bdiag, nimg;
};

// VTOK = true: V arrives token-major -- the layout every conv / GEMM epilogue writes -- and phase 2 builds its channel-row fragments with ds_read_b64_tr_b16
// (gfx950's transposing LDS read, as conv_wgrad_kernel.h: the 16 lanes of a group pass the addresses of a [4 keys][16 channels] block and lane i receives
// column i).  This is what lets the folded AttnBlock (blocks.hip: run_attn) use the normalised input itself as K AND as V: no k / v projections, no V^T tensor.
// LDS image of a 32-key chunk: row = key (Cp x 2 bytes, a multiple of 256), the 32-byte segment b (16 channels) of row r at segment position
// (b & ~7) | ((b + f(r)) & 7), f(r) = (r & 3) + 4 ((r >> 3) & 1): the 8 rows one LDS cycle serves (r .. r + 3 and r + 8 .. r + 11, two lane groups)
// fall on 8 different bank octets.  The k order inside an MFMA is the key order either way: same products, same sums as the V^T form.
// PROJ = true (C <= 512): proj_out (models/unet.py:189-191: 1x1 conv over the attention output, + bias, + the block's input) runs as a third phase on
// the workgroup's own 64 queries -- Y^T[co][q] = W_p . O^T over the channels in chunks of 32, W_p streamed through the V^T ring like V^T was, O kept in
// LDS as the bf16 image the GEMM would have read from HBM -- and leaves through conv_epilogue: the accumulator layout is the conv kernels' (a lane
// holds 4 consecutive output channels of one query = pixel), a query block is one 64-row statistics slab of the 16 x 16 map.  Same MFMA sequence per
// output and same epilogue as the stand-alone GEMM, hence the same bits; one 22 us launch and the O round trip through HBM gone per AttnBlock.
// QPROJ = true (round 5; PROJ and VTOK, i.e. the folded AttnBlock with C <= 512): the query projection q' = (Wk^T Wq) h + Wk^T bq -- the one GEMM the folded block still
// launched -- runs as phase 0 on the workgroup's own 64 queries, exactly like proj_out runs as phase 3: the 64 rows of h in LDS as 32-channel planes, Mq streamed through a ring
// of three 32 KB chunks, the result rounded to 16 bits (as the GEMM stored it) into the [64-channel step][query row] image phase 1 reads its query fragments from.  Phase 1's
// stages then hold key rows only (32 KB instead of 40).  Same MFMA sequence per output and the same rounding as the stand-alone GEMM, hence the same bits: the whole AttnBlock
// behind its GroupNorm is ONE launch.
struct AttnFusedCfg {
    static constexpr int N = 256, QB = 64, NTHREADS = 512;
    static constexpr int ST1 = (N + QB) * 128;                 // 40 KB: K rows then Q rows of one 64-channel step
    static constexpr int P_BYTES = QB * N * 2;                 // 32 KB
    static constexpr int MAX_CP = 512;                         // channels per phase-2 pass
    static constexpr int MAX_C = 2 * MAX_CP;
    static constexpr int ST2 = MAX_CP * 64;                    // 32 KB: rows of one pass x 32 keys
    static constexpr int RED_OFF = P_BYTES + 3 * ST2;          // 128 KB
    static constexpr int LDS_BYTES = RED_OFF + 2 * 4 * QB * 4;
    static_assert(3 * ST1 <= RED_OFF && LDS_BYTES <= 160 * 1024, "LDS");
};

template <bool PROJ, typename T_ = __bf16, bool VTOK = false, bool QPROJ = false>
__global__ __launch_bounds__(512, 2) void attn_fused_kernel(const AttnFusedArgs a, const ConvArgs pe) {
    static_assert(!QPROJ || (PROJ && VTOK), "the in-kernel query projection belongs to the folded AttnBlock with proj_out fused");
    using C = AttnFusedCfg;
    using T = T_;
    constexpr int N = C::N, QB = C::QB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16_mode_init<T>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The four query blocks of an image run on the SAME XCD (block id % 8, used for speed only), back to back: the image's K and V^T are
    // fetched into that XCD's L2 once and the other three workgroups hit it, instead of four XCDs each pulling them through the fabric.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int qb = seq & 3, b = (seq >> 2) * 8 + xcd;
    if (b >= a.B) return;
    if (a.bdiag && b * 4 + qb >= a.nimg) return;                      // (ragged last group: this query block is no image)
    const int Cc = a.C;

    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto make_q = [](const void* p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)p;
        return i32x4{(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    };
    const i32x4 q_q = make_q(a.q, a.q_bytes), q_k = make_q(a.k, a.k_bytes), q_vt = make_q(a.v, a.v_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const i32x4& rsrc, unsigned lds_addr, unsigned voff, int soff) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff)
                     : "memory");
    };

    // ---- phase-1 DMA geometry: 40 pieces of 8 rows x 128 B per step, five per wave; pieces 0-31 = key rows, 32-39 = query rows
    unsigned v1[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int piece = wave * 5 + j;
        const int row = piece * 8 + (lane >> 3);                       // row of the stage image (keys 0..255, then queries 0..63)
        const int u = (lane & 7) ^ ((row >> 1) & 7);
        const bool isk = row < N;
        const int tok = isk ? row : qb * QB + (row - N);
        v1[j] = (unsigned)(((long long)b * N + tok) * (isk ? a.k_ld : a.q_ld) * 2 + u * 16);
    }
    constexpr int ST1Q = N * 128;                                     // QPROJ: a phase-1 stage holds the 256 key rows only (32 KB); q' lives at Q_OFF
    constexpr int Q_OFF = 3 * ST1Q;                                   // 96 KB: eight [64 query rows][128 B] planes = 64 KB
    if constexpr (QPROJ) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 8 + (lane >> 3);          // key row
            const int u = (lane & 7) ^ ((row >> 1) & 7);
            v1[j] = (unsigned)(((long long)b * N + row) * a.k_ld * 2 + u * 16);
        }
    }
    auto issue1 = [&](int step, int buf) __attribute__((always_inline)) {
        if constexpr (QPROJ) {
            const unsigned base = lds0 + buf * ST1Q + wave * (4 * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16(q_k, base + j * 1024, v1[j], step * 128);
        } else {
        const unsigned base = lds0 + buf * C::ST1 + wave * (5 * 1024);
#pragma unroll
        for (int j = 0; j < 5; ++j) dma16(wave * 5 + j < N / 8 ? q_k : q_q, base + j * 1024, v1[j], step * 128);          // (wave-uniform choice)
        }
    };
    // ---- phase-2 DMA geometry: Cp / 16 pieces of 16 rows x 64 B per 32-key chunk (conv kernel's rotated 64-byte rows); Cp = channels per pass
    const int npass = Cc > C::MAX_CP ? 2 : 1;
    const int Cp = Cc / npass;
    const int p2 = Cp / 16;                                            // pieces per chunk (<= 32)
    const bool own_rows = Cp == 512;                                   // every wave fetches exactly the 64 rows it reads (phases 2 and 3): no per-step workgroup barrier there
    const int un2 = (lane & 3) ^ ((lane >> 3) & 2);
    const int rowb = Cp * 2;                                           // VTOK: bytes per key row of a chunk
    unsigned v2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j;
        if constexpr (VTOK) {
            // the chunk image is filled linearly, 1 KB per piece: this lane's 16 bytes are slot s16 of row r; it fetches the unit that belongs there
            const int off = piece * 1024 + lane * 16;
            const int r = off / rowb, s16 = (off - r * rowb) >> 4;
            const int sp = s16 >> 1, f = (r & 3) + 4 * ((r >> 3) & 1);
            const int bseg = (sp & ~7) | ((sp - f) & 7);
            v2[j] = piece < p2 ? (unsigned)((((long long)b * N + r) * a.v_ld) * 2 + (bseg * 2 + (s16 & 1)) * 16) : 0xFFFF0000u;
        } else {
            const int row = piece * 16 + (lane >> 2);                  // channel of the pass
            v2[j] = piece < p2 ? (unsigned)((((long long)b * Cc + row) * N) * 2 + un2 * 16) : 0xFFFF0000u;
        }
    }
    auto issue2 = [&](int pass, int chunk, int buf) __attribute__((always_inline)) {
        const unsigned base = lds0 + C::P_BYTES + buf * C::ST2 + wave * (4 * 1024);
        const int soff = VTOK ? (chunk * 32 * a.v_ld + pass * Cp) * 2 : pass * Cp * N * 2 + chunk * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(q_vt, base + j * 1024, v2[j], soff);
    };

    // =========================== phase 1: S^T = K . Q^T ===========================
    const int wm = wave >> 1, wn = wave & 1;                           // key block of 64, query block of 32
    const int sw = (lane >> 1) & 7, ku = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int slot = (ks * 4 + ku) ^ sw;
        a_off[ks] = (wm * 64 + (lane & 15)) * 128 + slot * 16;                   // key rows
        b_off[ks] = N * 128 + (wn * 32 + (lane & 15)) * 128 + slot * 16;         // query rows
    }
    f32x4 s_acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s_acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = Cc / 64;
    if constexpr (QPROJ) {
        // =========================== phase 0: q'^T = Mq . h_q^T (+ cq), 16-bit, into the query image of phase 1 ===========================
        // LDS: h_q as Cc / 32 planes of [64 query rows][64 B] (the conv kernels' rotated rows) at 0 (<= 64 KB), Mq chunks (Cc rows x 32 channels = 32 KB) in a ring of
        // three behind it; waves = 8 blocks of 64 output channels; a wave reads only the Mq rows it fetched itself when C = 512 (own_rows), as in phase 3
        constexpr int HQ_BYTES = QB * 512 * 2, ST0 = 512 * 64;
        static_assert(HQ_BYTES + 3 * ST0 <= 160 * 1024 && Q_OFF + QB * 512 * 2 <= 160 * 1024, "LDS");
        const i32x4 q_mw = make_q(a.qw, a.qw_bytes);
        const int np0 = Cc / 32;                                       // planes of h_q
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                  // 4 pieces of 16 rows per plane: piece = plane * 4 + row block
            const int piece = wave * 8 + j;
            const int pl = piece >> 2, row = (piece & 3) * 16 + (lane >> 2);
            if (pl < np0) dma16(q_k, lds0 + piece * 1024, (unsigned)(((long long)b * N + qb * QB + row) * a.k_ld * 2 + un2 * 16), pl * 64);
        }
        unsigned v0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 16 + (lane >> 2);         // output channel
            // slab-major copy: a 1 KB piece (16 rows x 64 B) is one contiguous run of whole cache lines; the plain matrix gives 16 half lines a kilobyte apart
            v0[j] = row < Cc ? (unsigned)((a.qw_slab ? row * 32 : row * a.qw_ld) * 2 + un2 * 16) : 0xFFFF0000u;
        }
        const int qstep = a.qw_slab ? a.qw_slab * 2 : 64;             // bytes between 32-channel chunks
        auto issue0 = [&](int chunk, int buf) __attribute__((always_inline)) {
            const unsigned base = lds0 + HQ_BYTES + buf * ST0 + wave * (4 * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16(q_mw, base + j * 1024, v0[j], chunk * qstep);
        };
        issue0(0, 0);
        issue0(1, 1);
        f32x4 y_acc[4][4];                                             // [query block][channel block]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) y_acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ku0 = lane >> 4;
        const int wa_off = HQ_BYTES + lds_off(wave * 64 + (lane & 15), ku0);
        const int hb_off = lds_off(lane & 15, ku0);
        const bool own0 = Cc == 512;
        int buf = 0;
        for (int k = 0; k < np0; ++k) {
            if (k + 1 < np0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (k == 0 || !own0) __builtin_amdgcn_s_barrier();         // k = 0: every wave's pieces of h_q are in
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < np0) issue0(k + 2, buf >= 1 ? buf - 1 : 2);
            const char* wb = smem + buf * ST0;
            const char* hb = smem + k * (QB * 64);
            uint4 wf[4], hf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const uint4*)(wb + wa_off + j * (16 * 64));
#pragma unroll
            for (int i = 0; i < 4; ++i) hf[i] = *(const uint4*)(hb + hb_off + i * (16 * 64));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16t<T>(y_acc[i][j], hf[i], wf[j]);
            buf = buf == 2 ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // every wave is done with h_q and the ring
        __builtin_amdgcn_sched_barrier(0);
        issue1(0, 0);                                                  // the first key stages travel while q' is written
        if (nk > 1) issue1(1, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c0 = wave * 64 + j * 16 + (lane >> 4) * 4;       // four consecutive output channels
            const float4 cb = (a.qbias != nullptr && c0 < Cc) ? *(const float4*)(a.qbias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int plane = c0 >> 6, unit = (c0 & 63) >> 3, half = (c0 >> 2) & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qn = i * 16 + (lane & 15);
                // (the GEMM's epilogue: fma(acc, 1, bias), one rounding)
                const uint2 qv = make_uint2(TI<T>::pack2(__builtin_fmaf(y_acc[i][j][0], 1.0f, cb.x), __builtin_fmaf(y_acc[i][j][1], 1.0f, cb.y)),
                                            TI<T>::pack2(__builtin_fmaf(y_acc[i][j][2], 1.0f, cb.z), __builtin_fmaf(y_acc[i][j][3], 1.0f, cb.w)));
                if (c0 < Cc) *(uint2*)(smem + Q_OFF + plane * (QB * 128) + qn * 128 + ((unit ^ ((qn >> 1) & 7)) << 4) + half * 8) = qv;
            }
        }
    } else {
    issue1(0, 0);
    if (nk > 1) issue1(1, 1);
    }
    {
        int buf = 0;
        for (int k = 0; k < nk; ++k) {
            if (k + 1 < nk) { if constexpr (QPROJ) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < nk) issue1(k + 2, buf >= 1 ? buf - 1 : 2);
            const char* base = smem + buf * (QPROJ ? ST1Q : C::ST1);
            const char* qbase = QPROJ ? smem + Q_OFF + k * (QB * 128) - N * 128 : base;      // (b_off carries the N * 128 of the stage image)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 af[4], bf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const uint4*)(base + a_off[ks] + i * (16 * 128));
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *(const uint4*)(qbase + b_off[ks] + j * (16 * 128));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma16<T>(s_acc[i][j], af[i], bf[j]);
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // every wave is done with the phase-1 stages
    __builtin_amdgcn_sched_barrier(0);
    issue2(0, 0, 0);                                                   // the first V^T chunks travel while the softmax runs
    issue2(0, 1, 1);

    // =========================== softmax over the keys, per query ===========================
    // lane: query column wn*32 + j*16 + (lane & 15); keys wm*64 + i*16 + (lane >> 4)*4 + r
    float* red = (float*)(smem + C::RED_OFF);                          // [2][4 key blocks][64 queries]
    const float sl2 = a.alpha * 1.4426950408889634f;                   // scores in log2 units
    float mx[2], sm[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s_acc[i][j][r] = (a.bdiag && wm != qb) ? -INFINITY : s_acc[i][j][r] * sl2; m = fmaxf(m, s_acc[i][j][r]); }      // (wm = this wave's key block)
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < 16) red[wm * QB + wn * 32 + j * 16 + lane] = m;
        mx[j] = m;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qn = wn * 32 + j * 16 + (lane & 15);
        const float m = fmaxf(fmaxf(red[qn], red[QB + qn]), fmaxf(red[2 * QB + qn], red[3 * QB + qn]));
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s_acc[i][j][r] - m); s_acc[i][j][r] = e; s += e; }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane < 16) red[4 * QB + wm * QB + qn] = s;
        sm[j] = s;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // P[query][key] bf16 into LDS: row = query (512 B), 16-byte unit u (8 keys) in slot u ^ (row & 15); this lane owns 4 consecutive keys
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qn = wn * 32 + j * 16 + (lane & 15);
        const float* r4 = red + 4 * QB;
        const float inv = 1.0f / (r4[qn] + r4[QB + qn] + r4[2 * QB + qn] + r4[3 * QB + qn]);       // fixed order: identical in every wave
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key0 = wm * 64 + i * 16 + (lane >> 4) * 4;
            const int unit = key0 >> 3, half = (key0 >> 2) & 1;
            const unsigned lo = TI<T>::pack2(s_acc[i][j][0] * inv, s_acc[i][j][1] * inv), hi = TI<T>::pack2(s_acc[i][j][2] * inv, s_acc[i][j][3] * inv);
            *(uint2*)(smem + qn * 512 + ((unit ^ (qn & 15)) << 4) + half * 8) = make_uint2(lo, hi);
        }
    }
    (void)mx; (void)sm;

    // =========================== phase 2: O^T = V^T . P^T ===========================
    // wave = block of Cp/8 channels of the pass; fragments: A = V^T rows (channels), B = P rows (queries); K = 32 keys per chunk
    const int cw = Cp / 8;                                             // channels per wave and pass: 16 ... 64
    const int nfi = cw / 16;                                           // channel fragments per wave (Cp multiple of 128)
    const int va_off = C::P_BYTES + lds_off(wave * cw + (lane & 15), ku);                 // + i * 16 rows * 64 B
    unsigned va_tr[4];                                                 // VTOK: address of this lane's 8 bytes of the [4 keys][16 channels] block, keys (lane >> 4) * 8 ... + 3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int li = lane & 15, g = lane >> 4, bs = wave * (cw >> 4) + i;
        const int fr = (li >> 2) + 4 * (g & 1);                        // f(row) of row g * 8 + (li >> 2), and of the row four below it
        va_tr[i] = lds0 + C::P_BYTES + (g * 8 + (li >> 2)) * rowb + ((bs & ~7) | ((bs + fr) & 7)) * 32 + (li & 3) * 8;
    }
    int pb_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pb_off[j] = (j * 16 + (lane & 15)) * 512;                // + ((chunk*4 + ku) ^ (row & 15)) << 4
    const int prow15 = lane & 15;
    T* op = (T*)a.o + ((long long)b * N + qb * QB) * Cc;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass > 0) {                                                // every wave is done with the ring: restart it on the upper half of the channels
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue2(pass, 0, 0);
            issue2(pass, 1, 1);
        }
        f32x4 o_acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) o_acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int buf = 0;
        for (int k = 0; k < N / 32; ++k) {
            if (k + 1 < N / 32) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            // own_rows (Cp = 512): a wave's V^T rows -- the channels it computes -- are exactly the rows it fetched itself (pieces 4 wave ... 4 wave + 3), and the ring
            // slot it overwrites holds only rows it alone reads: its own counted wait is all the synchronisation a step needs.  Only k = 0 needs the workgroup (P)
            if (k == 0 || !own_rows || VTOK) __builtin_amdgcn_s_barrier();       // (VTOK: a wave fetches key rows and reads channel columns)
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < N / 32) issue2(pass, k + 2, buf >= 1 ? buf - 1 : 2);
            const char* vb = smem + buf * C::ST2;
            uint4 af[4], bf[4];
            const int pslot = ((k * 4 + ku) ^ prow15) << 4;
            if constexpr (VTOK) {
                unsigned long long lo[4], hi[4];
                const unsigned sb = buf * C::ST2;
                // NO control flow between a transposing read and the wait behind it: the read fills its register asynchronously, and at a merge (`if (i < nfi) read; else
                // zero`) the compiler may copy a register that has not been filled yet -- measured in round 5: C = 1024 in f16 turned non-deterministic (1e-2 off) when an
                // unrelated line of the softmax changed the schedule.  Fragments i >= nfi re-read fragment 0 and are never multiplied.
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned ad = (i < nfi ? va_tr[i] : va_tr[0]) + sb;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[i]) : "v"(ad) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[i]) : "v"(ad + 4 * rowb) : "memory");
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const uint4*)(smem + pb_off[j] + pslot);
                // the transposing reads are invisible to the compiler's wait-count bookkeeping: everything has landed behind this wait
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]) :: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = uint4{(unsigned)lo[i], (unsigned)(lo[i] >> 32), (unsigned)hi[i], (unsigned)(hi[i] >> 32)};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < nfi) af[i] = *(const uint4*)(vb + va_off + i * (16 * 64));
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const uint4*)(smem + pb_off[j] + pslot);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nfi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16<T>(o_acc[i][j], af[i], bf[j]);
            buf = buf == 2 ? 0 : buf + 1;
        }
        // ---- epilogue of the pass: lane holds 4 consecutive channels (rows of the fragment) of query j*16 + (lane & 15)
        if constexpr (PROJ) {
            // every wave is done with P and the ring: the first two W_p chunks go out, and O (bf16, as the GEMM would have read it) goes to LDS as
            // 16 planes of [64 queries][32 channels] in the conv kernels' rotated 64-byte rows
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nfi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qn = j * 16 + (lane & 15);
                    const int c0 = pass * Cp + wave * cw + i * 16 + (lane >> 4) * 4;
                    const float4 vb = a.vbias ? *(const float4*)(a.vbias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const uint2 ov = make_uint2(TI<T>::pack2(o_acc[i][j][0] + vb.x, o_acc[i][j][1] + vb.y), TI<T>::pack2(o_acc[i][j][2] + vb.z, o_acc[i][j][3] + vb.w));
                    if constexpr (PROJ) *(uint2*)(smem + (c0 >> 5) * (QB * 64) + lds_off(qn, (c0 & 31) >> 3) + ((c0 >> 2) & 1) * 8) = ov;
                    else *(uint2*)(op + (long long)qn * Cc + c0) = ov;
                }
    }
    if constexpr (PROJ) {
        // =========================== phase 3: Y^T = W_p . O^T, then the conv epilogue ===========================
        constexpr int O_BYTES = QB * 512 * 2;                              // 64 KB: the image of 512 channels (C <= 512: host check)
        constexpr int ST3 = 512 * 64;                                      // 32 KB: W_p rows (output channels) x 32 input channels
        static_assert(O_BYTES + 3 * ST3 <= 160 * 1024, "LDS");
        const bool sm3 = pe.w_sm != nullptr;                              // the slab-major copy [C / 32][w_rows][32] of Wp Wv (blocks.hip: run_attn)
        const i32x4 q_w = make_q(sm3 ? pe.w_sm : pe.w, pe.w_bytes);
        unsigned v3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = wave * 4 + j;
            const int row = piece * 16 + (lane >> 2);                      // output channel
            v3[j] = row < pe.w_rows && row < Cc ? (unsigned)((sm3 ? row * 32 : row * pe.w_row_stride) * 2 + un2 * 16) : 0xFFFF0000u;
        }
        const int wstep = sm3 ? pe.w_rows * 64 : 64;                      // bytes between 32-channel chunks
        auto issue3 = [&](int chunk, int buf) __attribute__((always_inline)) {
            const unsigned base = lds0 + O_BYTES + buf * ST3 + wave * (4 * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16(q_w, base + j * 1024, v3[j], chunk * wstep);
        };
        issue3(0, 0);
        issue3(1, 1);
        f32x4 y_acc[4][4];                                                 // [query block][channel block]: conv_epilogue's acc[i][j]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) y_acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int wa_off = O_BYTES + lds_off(wave * 64 + (lane & 15), ku);             // + j * 16 rows * 64 B: W_p rows of this wave's 64 output channels
        const int ob_off = lds_off(lane & 15, ku);                                      // + i * 16 rows * 64 B + chunk plane
        const int nch = Cc / 32;
        int buf = 0;
        for (int k = 0; k < nch; ++k) {
            if (k + 1 < nch) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (k == 0 || !own_rows) __builtin_amdgcn_s_barrier();          // k = 0: every wave's part of the O image is visible; afterwards a wave reads only the W_p rows it fetched itself
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < nch) issue3(k + 2, buf >= 1 ? buf - 1 : 2);
            const char* wb = smem + buf * ST3;
            const char* ob = smem + k * (QB * 64);
            uint4 wf[4], of[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const uint4*)(wb + wa_off + j * (16 * 64));
#pragma unroll
            for (int i = 0; i < 4; ++i) of[i] = *(const uint4*)(ob + ob_off + i * (16 * 64));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16t<T>(y_acc[i][j], of[i], wf[j]);       // weight fragment = row operand: [channel][query] result (conv_kernel.h)
            buf = buf == 2 ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // every wave is done with the operand images the epilogue tile overlays
        __builtin_amdgcn_sched_barrier(0);
        // the workgroup's 64 queries are rows 4 qb ... 4 qb + 3 of the image's 16 x 16 map: wave row `qb` of a 256-pixel tile; its 8 waves are 8 column blocks
        // (round 5: the 16-bit-tile epilogue takes the residual as well -- conv_kernel.h: conv_epilogue_packed --: same bits, a third of the fp32 tile's LDS traffic, none of its bank conflicts)
        conv_epilogue<T, 16, 16, 4, 4, 4, EpiNoHook, false, 1>(pe, y_acc, smem, true, wave, lane, qb, wave, b, 0, 0, 0, 0, 0, EpiNoHook(), false);
        gn_arrive<C::NTHREADS>(pe, b, 1, 256, (int*)smem, (int)threadIdx.x);       // four query blocks complete an image
    }
}

}  // namespace wdm
