// libccsm GRU layers 1-2 in split-mx arithmetic on the 16-wide matrix instructions: v_mfma_f32_16x16x32_f16 (main product) +
// v_mfma_scale_f32_16x16x128_f8f6f4 (both correction terms of TWO pairs of k-blocks per instruction).  Included by ccsm_api.hip after
// ccsm_gru_f3s.hip.  The arithmetic, the input / output formats, the LDS layout, the x ring and its transfers are ccsm_gru_mx.hip's
// (gru_layer12_mx_kernel), byte for byte; what changes is the matrix instruction and with it which values a weight fragment holds.
//
// Why (round 5: profiles/r05_m, r05_n, r05_b2, r05_i2): the GRU kernels run at the package power cap, where a kernel's time is its energy.
// The 16x16 shapes reduce 32 (fp16) / 128 (MX) k inside the array before they touch an accumulator - half the fp32 accumulate traffic
// per MAC - and the cap lets the split-mx instruction mix sustain 15 % more on them (1279 against 1111 TFLOP/s); the kernel's recurrent
// phase in isolation runs 9.7 % faster, its input-part phase 2.7 % (tools/ubench/phase_h_shapes.hip).
//
// What came of it (round 6, profiles/r06_a .. r06_h): parity-green at the first run in every form, and 10-12 % SLOWER than gru_layer12_mx_kernel
// on the same boxes (2.01 against 1.83 ms per 12288-site launch) - at 1360 W and 1.97 GHz, i.e. BELOW the power cap the 32-wide kernel sits on
// (1385 W, 1.80 GHz): the energy is saved and cannot be spent.  Four schedules were built (every request at the pair's end | three weight
// slots, requests in front of the barrier | pairs software-pipelined across their barriers | every request behind a few MFMAs of its own:
// this file) and all land at 91-97 k cycles per step against the 32-wide kernel's 74 k.  The stamps say why: the CU's vector-memory path takes
// one 1-KiB request per 16 cycles from all eight waves together (eleven requests per wave in a row: 1400 cycles for the last wave), a wave
// that waits for its slot issues no MFMA, and the two waves of a SIMD wait at the same time - one barrier per pair keeps all eight in one
// phase.  Per double pair the path is busy 2200 cycles (weights 49 + x 12 KiB per pair), the matrix pipe 1920: the kernel's time is their
// SUM, whatever the order of the requests.  The 32-wide kernel carries the same bytes and loses less of the overlap (half as many matrix
// instructions between the same requests).  Diagnostic builds (kMx16Diag, profiles/r06_d, r06_g): 0.67 ms of the 1.99 are phase B and the
// tail, 0.72 ms the input-part phases' MFMAs with every memory consumer removed, 0.60 ms the ring refills, the L2 weight stream, the blob
// reads and the barriers.  Kept behind CCSM_MX_SHAPE16=1 (ccsm_create), tested, gated (tools/isa_gate.py); not the default.
//
// Operands.
//   * B (activations): x_t, the state and the layer output stay 32-row fragments [hi | blob] (ccsm_gru_mx.hip).  A B operand of the fp16
//     instruction (lane (n', q) <- row n', k = 8 q + j of 32) for rows [16 h, 16 h + 16) of a 32-row tile and the pair of k-blocks
//     (2 P, 2 P + 1) is those fragments read with per-lane addresses: lane (n', q) reads 16 B at fragment(2 P + (q >> 1)) +
//     (16 h + n' + 32 (q & 1)) * 16 (ccsm_gru_f3s.hip).  The scaled instruction contracts K = 128 = four blocks of 32: lane (n', q) of
//     its B operand is the existing activation blob (lane (n, g) of a 32-row blob fragment = 32 values of row n: g = 0 x_hi, g = 1 x_lo,
//     kMxPerm order) of pair q >> 1 of the DOUBLE pair at lane position 16 h + n' + 32 (q & 1): bytes 0-15 from the corr fragment of the
//     pair's first k-block, 16-23 from the second's.
//   * A (weights): lane (m, q) of a hi fragment <- unit-tile row m, k = 32 pair + 8 q + j; lane (m, q) of a blob <- 32 values (kMxPerm
//     order) of unit-tile row m of pair 2 D + (q >> 1), term q & 1 (0: W_lo 2^11, 1: W_hi), one E8M0 byte per lane and blob.  Unit tiles as
//     in ccsm_gru_f3s.hip: tile T, row 4 q + i <-> hidden unit 32 wave + 8 q + 4 T + i, so the step tail's hi fragments need no lane
//     exchange; the blobs (32 units of ONE row per lane) do: the four lane groups exchange 4-dword chunks (eight v_permlane32_swap +
//     eight v_permlane16_swap per row tile; the 32-wide tail: eight swaps).
//   * The x ring and the double pair (tools/ubench/phase_h_shapes.hip, px16): the correction product of pairs (P - 1, P) needs both pairs'
//     blobs and the ring releases a slot at its pair's barrier - so the lanes that take pair P - 1's blob (q < 2: lanes 0-31) read it WHILE
//     PAIR P - 1 IS IN ITS SLOT and keep it in registers across the barrier; at pair P lanes 32-63 read theirs into the same registers
//     (exec-masked ds_reads), then the double pair's correction instructions issue.  Ring, refills and barriers stay as they are.
//   wst : per (direction, wave) one stream in consumption order (1 KiB fragments: lane * 16; scale dwords: lane * 4)
//           phase A (r, z input part), per double pair DA (8): hi of pair 2 DA (T, g) at (2 T + g) KiB | hi of pair 2 DA + 1 at 4 + ... |
//                    fp4 blobs (T, g) at 8 + (2 T + g) KiB | scale dwords at 12 KiB, byte 2 T + g                         12.25 KiB
//           phase B (recurrent part), per double pair D (4): hi of pair 2 D (T, g) at (3 T + g) KiB | hi of pair 2 D + 1 at 6 + ... |
//                    fp4 blobs (T, g) at 12 + (3 T + g) KiB | scale dwords of T = 0 at 18 KiB, of T = 1 at 18 KiB + 256, byte g 18.5 KiB
//           phase C (n input part), per double pair DC (8) of consumption POSITIONS (zig-zag: position pp <-> pair 15 - pp): hi of
//                    position 2 DC (T) at T KiB | hi of position 2 DC + 1 at 2 + T KiB | fp6 blobs (T): bytes 0-15 at (4 + T) KiB,
//                    bytes 16-23 at 6 KiB + T * 512 (lane * 8) | scale dwords at 7 KiB, byte T                              7.25 KiB
//   bias: natural unit order [direction][wave][set r, z, n_x, n_h][32] (pack_bias_natural)
//   LDS : gru_layer12_mx_kernel's (h fragments [kb 16][bt][hi | corr] 96 KiB | x ring 4 x 12 KiB | fp8 residuals 12 KiB | biases 4 KiB); the
//         fp8 residuals (x 2^16) of a lane's own 8 units: row half 0 in bytes 8-15 of its slot in the corr fragment of the wave's second
//         k-block, half 1 at LO_OFF
// Counted waits: a wave's vector-memory LOADS retire in order, so "my part of the next pair's transfer has landed" is s_waitcnt vmcnt(N)
// with N = the loads the wave has issued behind that transfer.  The N of every pair comes from ONE constexpr model of the step's request
// sequence (mx16_waits below: the same per-site counts the code issues) instead of hand-counted immediates; stores are not counted (they
// may retire ahead of older loads: a smaller N only waits longer), and tools/isa_gate.py re-derives every N from the code object.
#include <hip/hip_runtime.h>

namespace ccsm {

typedef unsigned u32x6_t __attribute__((ext_vector_type(6)));

constexpr int kMx16PA = 12 * 1024 + 256;                                     // weight bytes of a phase-A double pair
constexpr int kMx16PB = 18 * 1024 + 512;                                     //                   phase-B
constexpr int kMx16PC = 7 * 1024 + 256;                                      //                   phase-C
constexpr int kMx16OffB = (kKB12 / 4) * kMx16PA;
constexpr int kMx16OffC = kMx16OffB + (kKBH / 4) * kMx16PB;
constexpr int kMx16WBytes = kMx16OffC + (kKB12 / 4) * kMx16PC;               // 230 KiB per (direction, wave)

#define CCSM_FENCE asm volatile("" ::: "memory")

// Diagnostic builds (results wrong on purpose; tools/ab_build.sh <name> -DCCSM_MX16_DIAG=<bits>): what one consumer of the input-part phases
// costs in cycles and joules.  1: no correction products in phases A and C; 2: no blob reads from LDS; 4: no barriers in phases A and C;
// 8: every weight request reads one of the stream's first four fragments (L1-resident: same requests, no L2 -> CU traffic); 16: no ring refills; 32: no counted waits; 64: no main products in phases A and C;
// 128: gate math (sigmoids, tanh) replaced by copies
#ifndef CCSM_MX16_DIAG
#define CCSM_MX16_DIAG 0
#endif
constexpr int kMx16Diag = CCSM_MX16_DIAG;

// ---- vector-memory requests per site of a step (the code below issues exactly these; mx16_waits() counts with the same functions).
// An ITERATION i of an input-part phase is what lies between barrier i - 1 and barrier i: the second row half of pair i - 1's main products,
// the requests that refill its weight slot, the ring refill, (even i) the double pair's correction products, then pair i's first row half.
// The CU's vector-memory path takes one 1-KiB request per 16 cycles from all eight waves together and a wave stalls while its request waits
// for a slot: eleven requests in a row held the second wave of a SIMD for 1400 cycles (profiles/r06_h).  So every request sits behind a few
// MFMAs of its own: one behind each weight fragment's last use, one behind each group of correction products.
constexpr int kMx16NPair = kKB12 / 2;                                        // 16 pairs of k-blocks of input
constexpr int kMx16SA = 3;                                                   // phase-A pair slots of hi fragments (pair P lives in slot P % 3)
constexpr bool mx16_has_blob(int i) { return i >= 2 && !(i & 1) && (i >> 1) < kMx16NPair / 2; }           // iteration i = 2 D + 2 requests double pair D + 1's blobs + scales
// phase-A iteration i (1 .. 16), behind fragment f (= 2 T + g) of pair i - 1's second half: fragment f of pair i + 2 into the slot just read
// (iterations 15, 16: phase B's first two sets, six fragments over the four points)
constexpr int mx16_req_a(int i, int f) { return i + 2 < kMx16NPair ? 1 : i >= kMx16NPair - 1 ? ((f & 1) ? 1 : 2) : 0; }
constexpr int mx16_req_bset(int u) { return 6 + (u % 3 == 2 ? 2 : 0); }      // set u of phase B (u % 3 = 2: blobs + the two scale dwords)
constexpr int kMx16ReqBEnd = 4;                                              // behind use 10 of phase B: hi of phase-C positions 0, 1
constexpr int kMx16ReqR = 4 + 5;                                             // behind r = sigmoid(R): hi of positions 2, 3, blobs + scales of double pair 0
constexpr int kMx16ReqTail = 4 + 5;                                          // behind the tail: hi of the next step's pair 2, blobs + scales of its double pair 0

// Per consumption c (0-15: phase-A pairs, 16-31: phase-C positions) the wait in front of its barrier wants the transfer of consumption c + 1,
// issued behind the barrier of consumption c + 1 - RS (phase-A pair 15's refill: behind phase B).  A transfer is one request of every wave
// that moves a fragment (`a`) and a second one of waves 0-3 (`b`: twelve fragments for eight waves at 96 rows); lo[c] / hi[c] = the requests a
// wave with one / two issues behind its last one of the awaited transfer.
struct Mx16Waits { int lo[2 * kMx16NPair]; int hi[2 * kMx16NPair]; };
constexpr Mx16Waits mx16_waits(int rs) {
    constexpr int NP = kMx16NPair, NC = 2 * NP, CAP = 12 * NC + 32;
    int kind[CAP] = {}, val[CAP] = {};                                       // 0: val weight requests; 1: transfer request `a` of consumption val; 3: request `b`; 2: the wait of consumption val
    int n = 0;
    for (int ph = 0; ph < 2; ++ph) {                                         // phase A, then (behind phase B) phase C: the same iteration structure
        for (int i = 0; i <= NP; ++i) {                                      // iterations 0 .. 16 (16 = the epilogue behind barrier 15)
            const bool refill = i >= 1 && (ph == 1 || i < NP);               // (phase A's last refill is deferred)
            const int cprev = ph * NP + i - 1;
            if (i >= 1) {
                if (ph == 0) for (int f = 0; f < 4; ++f) { kind[n] = 0; val[n++] = mx16_req_a(i, f); }
                else for (int T = 0; T < 2; ++T) { kind[n] = 0; val[n++] = 1; }                          // phase C: hi (T) of position i + 3
                if (refill) { kind[n] = 1; val[n++] = (cprev + rs) % NC; }
                if (!(i & 1)) {                                              // behind each of the six groups of correction products
                    for (int j = 0; j < 6; ++j) {
                        if (mx16_has_blob(i) && j < 5) { kind[n] = 0; val[n++] = 1; }
                        if (refill && j == (mx16_has_blob(i) ? 5 : 0)) { kind[n] = 3; val[n++] = (cprev + rs) % NC; }
                    }
                } else if (refill) { kind[n] = 3; val[n++] = (cprev + rs) % NC; }                       // odd i: in the middle of pair i's first half
            }
            if (i < NP) { kind[n] = 2; val[n++] = ph * NP + i; }
        }
        if (ph == 0) {
            int b = kMx16ReqBEnd;
            for (int u = 2; u < 12; ++u) b += mx16_req_bset(u);
            kind[n] = 0; val[n++] = b;
            kind[n] = 1; val[n++] = (NP - 1 + rs) % NC;                      // the deferred refill, both requests
            kind[n] = 3; val[n++] = (NP - 1 + rs) % NC;
            kind[n] = 0; val[n++] = kMx16ReqR;
        }
    }
    kind[n] = 0; val[n++] = kMx16ReqTail;
    Mx16Waits r = {};
    for (int i = 0; i < n; ++i) {
        if (kind[i] != 2) continue;
        const int want = (val[i] + 1) % NC;
        int lo = 0, hi = 0, seen_b = 0;
        for (int j = (i + n - 1) % n; ; j = (j + n - 1) % n) {               // backwards (cyclically: the previous step) to the awaited transfer
            if (kind[j] == 3 && val[j] == want) { seen_b = 1; continue; }    // its `b`: the end of what a two-request wave counts
            if (kind[j] == 1 && val[j] == want) break;
            if (kind[j] == 2) continue;                                      // (another pair's wait: not a request)
            const int c = kind[j] == 0 ? val[j] : 1;
            if (kind[j] != 3) lo += c;                                       // (a one-request wave issues no `b` at all)
            if (!seen_b) hi += c;
        }
        r.lo[val[i]] = lo;
        r.hi[val[i]] = hi;
    }
    return r;
}

// fp16 product with the accumulator tied: ccsm_gru_f3s.hip's mfma32k / mfma32k_first / mfma_drain.
// Scaled product (A fp4, B fp6) with the accumulator tied; BYTE = which byte of the A scale dword.  (s_nop 1: hipcc assembles the six-register B
// operand with v_mov right in front of the statement and pads no hazard whose consumer is inside an asm string)
template <int BYTE>
__device__ __forceinline__ f32x4 mfma_corr16(uint4 a, uint32_t sa, u32x6_t bv, uint32_t sb, f32x4 c) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a);
    if constexpr (BYTE == 0) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else if constexpr (BYTE == 1) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,0,0] op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else if constexpr (BYTE == 2) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    return c;
}
// the same with an fp6 weight blob (the n gate's input part: ccsm_gru_mx.hip, kMxWFmtX)
template <int BYTE>
__device__ __forceinline__ f32x4 mfma_corr16_a6(u32x6_t av, uint32_t sa, u32x6_t bv, uint32_t sb, f32x4 c) {
    static_assert(BYTE == 0 || BYTE == 1, "one scale byte per unit tile");
    if constexpr (BYTE == 0) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,0,0] op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    return c;
}
__device__ __forceinline__ u32x6_t blob6(uint4 b0, uint2 b1) { return u32x6_t{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y}; }
// bytes 16-23 of a blob from LDS as ONE ds_read_b64 into its place in the operand: a plain load gets merged with its neighbour's into a
// ds_read2*_b64 whose 4-register result is then copied apart behind an s_waitcnt lgkmcnt(0) - two full drains of the LDS queue per pair
// (volatile, through an LDS-address-space pointer: a generic volatile access trips the backend's src_shared_base check - see gru_layer0_mx_kernel)
__device__ __forceinline__ uint2 rd8(const char* p) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef const volatile __attribute__((address_space(3))) u32x2* lds_p;
    const u32x2 v = *(lds_p)(__attribute__((address_space(3))) const char*)p;
    return make_uint2(v[0], v[1]);
}

__device__ __forceinline__ void swap16(uint32_t& x, uint32_t& y) {            // odd 16-lane rows of x <-> even rows of y
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// One row tile of new values in the 16x16 C layout: v[h][j] = unit 8 q + j of row 16 h + n' (lane (n', q)).  Produces
//   hi[h]      : this lane's 16 bytes of the hi fragments (k-block q >> 1 of the wave's pair, lane position n' + 32 (q & 1) + 16 h)
//   lo8[h]     : fp8 residuals (x 2^16) of the lane's own 8 values (private copy for the blend)
//   c0, c1     : THIS PHYSICAL LANE's 24 bytes of the row tile's activation blob fragment (lane L = n + 32 g: x_hi (g = 0) / x_lo (g = 1) of the
//                wave's 32 units of row n, kMxPerm order), gathered from the four lanes that hold a row
// CLAMP / scale: pack_pair_mx's (initial states: clamped, divided by kMxH0Div; GRU outputs: 0.25)
template <bool CLAMP>
__device__ __forceinline__ void mx16_pack(const float (&v)[2][8], float scale, uint4 (&hi)[2], uint2 (&lo8)[2], uint4& c0, uint2& c1) {
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    uint32_t ch[4][4];                                              // chunks by target lane group q_t = h + 2 g: [q_t][dword]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t hp[4];
        float lf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const half2p hh = {(_Float16)v[h][2 * j], (_Float16)v[h][2 * j + 1]};
            hp[j] = __builtin_bit_cast(uint32_t, hh);
            lf[2 * j] = v[h][2 * j] - (float)hh[0];
            lf[2 * j + 1] = v[h][2 * j + 1] - (float)hh[1];
            float l0 = lf[2 * j] * 4096.0f, l1 = lf[2 * j + 1] * 4096.0f;
            if constexpr (CLAMP) {
                const float top = 7.5f * scale;
                const half2p hc = {(_Float16)fminf(fmaxf(v[h][2 * j], -top), top), (_Float16)fminf(fmaxf(v[h][2 * j + 1], -top), top)};
                ch[h][j] = __builtin_bit_cast(uint32_t, hc);
                l0 = fminf(fmaxf(l0, -top), top);
                l1 = fminf(fmaxf(l1, -top), top);
            } else {
                ch[h][j] = hp[j];
            }
            ch[2 + h][j] = pack2((_Float16)l0, (_Float16)l1);
        }
        hi[h] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        uint32_t r[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(lf[4 * q + e], -kF8Clamp / kMxLoScale, kF8Clamp / kMxLoScale);
            short2v t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(short2v, hp[2 * q]), c[0], c[1], 1.0f / kMxLoScale, false);
            t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, c[2], c[3], 1.0f / kMxLoScale, true);
            r[q] = __builtin_bit_cast(uint32_t, t);
        }
        lo8[h] = make_uint2(r[0], r[1]);
    }
    // 4 x 4 exchange of the chunks between the lane groups: afterwards group q holds chunk q of all four source groups; the register a value
    // ends up in names its SOURCE group - stage 1 (bit 1: lanes 0-31 <-> 32-63) leaves sources 0, 1 in ch[0 / 1] and 2, 3 in ch[2 / 3], stage 2
    // (bit 0: odd <-> even rows of 16) sources 0, 2 in ch[0 / 2] and 1, 3 in ch[1 / 3]
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        swap32(ch[0][d], ch[2][d]);
        swap32(ch[1][d], ch[3][d]);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        swap16(ch[0][d], ch[1][d]);
        swap16(ch[2][d], ch[3][d]);
    }
    // natural dword 4 q_s + d = units 8 q_s + 2 d, + 1; blob dword p takes natural dword 4 ((p & 7) >> 1) + (p & 1) + 2 (p >> 3)  (kMxPerm)
    uint32_t p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nd = 4 * ((i & 7) >> 1) + (i & 1) + 2 * (i >> 3);
        p[i] = ch[nd >> 2][nd & 3];
    }
    blob_of(p, scale, c0, c1);
}

// DBG (only instantiated in a -DCCSM_PHASE_STAMPS build): workgroup 0 records the cycle counter at step start / behind phase A / B / C / the tail
template <bool OUT_FP8, int NB_ = kMxNB, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gru_layer12_mx16_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out, const uint4* __restrict__ wst,
                                                                   const float* __restrict__ bias, const float* __restrict__ h0, int rows_p,
                                                                   unsigned long long* __restrict__ dbg) {
    constexpr int NB = NB_, KX = kKB12, NPAIR = kMx16NPair, RS = kMxRS, SLOT_BYTES = mx_slot_bytes(NB);
    constexpr int X_OFF = mx12_xoff(NB), LO_OFF = mx12_looff(NB), BIAS_OFF = mx12_biasoff(NB);
    constexpr int PA = kMx16PA, PB = kMx16PB, PC = kMx16PC, OFF_B = kMx16OffB, OFF_C = kMx16OffC;
    constexpr Mx16Waits WT = mx16_waits(RS);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int lane16 = lane * 16;
    start_stagger((blockIdx.x >> 1) & 3);

    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(smem + BIAS_OFF)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];

    // per-lane offsets (bytes), each rebuilt from an opaque copy of lane * 16 where a phase needs it (ONE register live across the kernel instead
    // of five: the phases are at the register file's edge): lane position n' + 32 (q & 1) inside a fragment; + the k-block (q >> 1) of a pair
    // in a [kb][bt][hl] array of NB row tiles; + the PAIR (q >> 1) of a double pair of the state
    auto opq = [&](int v) -> int { asm volatile("" : "+v"(v)); return v; };
    auto f_lpos = [&]() -> int { const int l = opq(lane16); return ((l & 0x100) << 1) | (l & 0xf0); };
    auto f_lxs = [&]() -> int { const int l = opq(lane16); const int kbo = (l & 0x200) << 2; return (((l & 0x100) << 1) | (l & 0xf0)) + (NB == 1 ? kbo : NB == 2 ? (kbo << 1) : kbo + (kbo << 1)); };
    auto f_lbs = [&]() -> int { const int l = opq(lane16); return (((l & 0x100) << 1) | (l & 0xf0)) + (l >> 9) * (2 * NB * 2048); };
    const int own = wave * (2 * NB * 2 * 1024);                      // mx_hfrag(2 wave, 0, 0)

    // ---- h0 -> LDS: hi fragments, blobs (coarse scale) and residuals of this wave's own units, every row tile
    {
        const float* h0d = h0 + (size_t)dir * rows_p * kHidden;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            float v[2][8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* src = h0d + ((size_t)(tile0 + bt) * 32 + 16 * h + (lane & 15)) * kHidden + 32 * wave + 8 * (lane >> 4);
                const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                v[h][0] = a.x; v[h][1] = a.y; v[h][2] = a.z; v[h][3] = a.w; v[h][4] = b.x; v[h][5] = b.y; v[h][6] = b.z; v[h][7] = b.w;
            }
            uint4 hi[2], c0;
            uint2 lo8[2], c1;
            mx16_pack<true>(v, kMxH0Div, hi, lo8, c0, c1);
            *reinterpret_cast<uint4*>(smem + own + f_lxs() + bt * 2048) = hi[0];
            *reinterpret_cast<uint4*>(smem + own + f_lxs() + bt * 2048 + 256) = hi[1];
            *reinterpret_cast<uint4*>(smem + own + ((bt * 2 + 1) << 10) + lane16) = c0;
            *reinterpret_cast<uint4*>(smem + own + (((NB + bt) * 2 + 1) << 10) + lane16) = make_uint4(c1.x, c1.y, lo8[0].x, lo8[0].y);
            *reinterpret_cast<uint2*>(smem + LO_OFF + ((wave * NB + bt) * 64 + lane) * 8) = lo8[1];
        }
    }

    // ---- x transfers: fragment f = (kbl * NB + bt) * 2 + hl of a ring slot; wave w moves fragment w, waves 0-3 also w + 8 (NB = 3); the
    // descriptor starts at THIS workgroup's first tile (gru_layer12_mx_kernel)
    const u32x4_t xrs = dma_rsrc(xin + (size_t)tile0 * kSeqLen * KX * 2 * kFragU4);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)X_OFF;    // (cast first: see gru_layer0_mx_kernel)
    auto dma_pair = [&](int slot, int sd, int jd, int part = 2) {   // wave-uniform: ring slot, step (clamped), pair of x_t(sd); part 0 / 1: request `a` / `b` only
        const int sc_ = sd < kSeqLen ? sd : kSeqLen - 1;
        const int td = dir ? kSeqLen - 1 - sc_ : sc_;
        auto one = [&](int f) {
            const int hl = f & 1, bt = (f >> 1) % NB, kbl = (f >> 1) / NB;
            const int soff = ((((bt * kSeqLen + td) * KX + (2 * jd + kbl)) * 2 + hl) << 10);
            dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT_BYTES + (f << 10))));
        };
        if (part != 1) { if (4 * NB >= kWaves || wave < 4 * NB) one(wave); }             // `a`: 4 NB fragments per pair (NB = 1: waves 4-7 move nothing)
        if (part != 0) { if (4 * NB > kWaves && wave < 4 * NB - kWaves) one(wave + 8); }   // `b`
    };
    // the transfer of consumption jj + RS of step s (jj: 0-15 phase-A pairs, 16-31 phase-C positions, zig-zag) goes into the slot consumption jj
    // just vacated
    auto dma_ahead = [&](int slot, int s, int jj, int part = 2) {
        if constexpr (kMx16Diag & 16) return;
        const int g = jj + RS;
        const int c = g & (2 * NPAIR - 1);
        dma_pair(slot, s + (g >> 5), kMxZigZag && c >= NPAIR ? 2 * NPAIR - 1 - c : c & (NPAIR - 1), part);
    };
    // this wave's part of the next pair's transfer has landed: at most W weight requests + K transfers (one or two requests each) are younger
    auto xfer_wait = [&](auto cc) {
        constexpr int C = decltype(cc)::value;
        if constexpr (kMx16Diag & 32) return;
#ifdef CCSM_MX16_WAIT0          // debugging aid: every counted wait drains (a parity failure that this build does not show is a wrong count)
        wait_vm<0>();
#else
#ifdef CCSM_MX16_BAD_WAIT       // deliberately broken build (one request too many allowed in flight at pair 5): tools/isa_gate.py must reject it
        constexpr int OFF1 = C == 5 ? 1 : 0;
#else
        constexpr int OFF1 = 0;
#endif
        constexpr int NHI = WT.hi[C] + OFF1 > 63 ? 63 : WT.hi[C] + OFF1, NLO = WT.lo[C] + OFF1 > 63 ? 63 : WT.lo[C] + OFF1;
        if (4 * NB > kWaves && wave < 4 * NB - kWaves) wait_vm<NHI>();
        else if (4 * NB >= kWaves || wave < 4 * NB) wait_vm<NLO>();
#endif
    };

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * kMx16WBytes);
    const int bias_off = BIAS_OFF + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, (kMx16Diag & 8) ? (off & 0xc00) : off); };
    auto ws_at = [&](int off) -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, opq(lane16) >> 2, off, 0); };
    auto w8_at = [&](int off) -> uint2 {                            // bytes 16-23 of an fp6 blob: lane * 8
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(wrs, opq(lane16) >> 1, off, 0);
        return make_uint2(v[0], v[1]);
    };

    // weight registers.  Phase A: hi (T, g) of two pairs, blobs (T, g) + scale dword of two double pairs; phase B: two slots of six fragments in
    // use order (hi of pair 2 D | hi of pair 2 D + 1 | blobs of D) + the scale dwords (T = 0, 1) of the blobs in flight; phase C: hi (T) of four
    // positions, fp6 blobs (T) + scale dword of two double pairs
    uint4 wh[kMx16SA][4], wb[2][4];
    uint32_t wsa[2];
    uint4 w[2][6];
    uint32_t wsc[2] = {0, 0};
    uint4 wch[4][2], wcb0[2][2];
    uint2 wcb1[2][2];
    uint32_t wcs[2];
    auto ldA_hi2 = [&](auto pc, auto tc) {                          // hi fragments (T, g = r, z) of phase-A pair p -> slot p % 3              (2 requests)
        constexpr int p = decltype(pc)::value, T = decltype(tc)::value;
        wh[p % kMx16SA][2 * T] = w_at((p >> 1) * PA + (((p & 1) * 4 + 2 * T) << 10));
        wh[p % kMx16SA][2 * T + 1] = w_at((p >> 1) * PA + (((p & 1) * 4 + 2 * T + 1) << 10));
    };
    auto ldA_blob = [&](auto dc) {                                  // blobs + scales of phase-A double pair DA -> slot DA & 1             (5 requests)
        constexpr int DA = decltype(dc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) wb[DA & 1][i] = w_at(DA * PA + ((8 + i) << 10));
        wsa[DA & 1] = ws_at(DA * PA + (12 << 10));
    };
    auto ld_set = [&](auto UC) {                                    // use U of phase B (12): D = U / 3, kind = U % 3, slot = U & 1: mx16_req_bset(U) requests
        constexpr int U = decltype(UC)::value;
        constexpr int D = U / 3, KIND = U % 3, SL = U & 1;
#pragma unroll
        for (int i = 0; i < 6; ++i) w[SL][i] = w_at(OFF_B + D * PB + ((KIND * 6 + i) << 10));
        if constexpr (KIND == 2) {
            wsc[0] = ws_at(OFF_B + D * PB + (18 << 10));
            wsc[1] = ws_at(OFF_B + D * PB + (18 << 10) + 256);
        }
    };
    auto ldC_hi = [&](auto pc) {                                    // hi (T) of phase-C position pp -> slot pp & 3                         (2 requests)
        constexpr int pp = decltype(pc)::value;
        wch[pp & 3][0] = w_at(OFF_C + (pp >> 1) * PC + (((pp & 1) * 2 + 0) << 10));
        wch[pp & 3][1] = w_at(OFF_C + (pp >> 1) * PC + (((pp & 1) * 2 + 1) << 10));
    };
    auto ldC_blob = [&](auto dc) {                                  // fp6 blobs (T) + scales of phase-C double pair DC -> slot DC & 1     (5 requests)
        constexpr int DC = decltype(dc)::value;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            wcb0[DC & 1][T] = w_at(OFF_C + DC * PC + ((4 + T) << 10));
            wcb1[DC & 1][T] = w8_at(OFF_C + DC * PC + (6 << 10) + T * 512);
        }
        wcs[DC & 1] = ws_at(OFF_C + DC * PC + (7 << 10));
    };

    // ---- prologue: the ring's first pairs, the first two pair slots and the first double pair's blobs
#pragma unroll
    for (int g = 0; g < RS; ++g) dma_pair(g, 0, g);
    static_for<0, kMx16SA>([&](auto pc) {
        ldA_hi2(pc, std::integral_constant<int, 0>{});
        ldA_hi2(pc, std::integral_constant<int, 1>{});
    });
    ldA_blob(std::integral_constant<int, 0>{});
    wait_vm<4 * kMx16SA + 5>();                                     // all ring transfers (older than the weight requests)
    __syncthreads();                                                // ring, h0 fragments and biases are in LDS

    int slot = 0;                                                   // ring slot of the next consumption (wave-uniform)
    for (int s = 0; s < kSeqLen; ++s) {
        const int t = dir ? (kSeqLen - 1 - s) : s;
        auto stamp = [&](int k) {
            if constexpr (DBG) {
                if (dbg != nullptr && blockIdx.x == 0 && lane == 0) dbg[(s * kWaves + wave) * 5 + k] = __builtin_readcyclecounter();
            }
        };
        stamp(0);
        // DBG: step 10 also records, per iteration of phases A (0-15) and C (16-31), seven stamps: behind the barrier / behind pair i - 1's second half /
        // behind the requests and the refill / behind the correction products / behind pair i's first half / behind the counted wait / (next: behind the barrier)
        auto pstamp = [&](int idx, int k) {
            if constexpr (DBG) {
                if (dbg != nullptr && s == 10 && blockIdx.x == 0 && lane == 0) dbg[kSeqLen * kWaves * 5 + (wave * 32 + idx) * 7 + k] = __builtin_readcyclecounter();
            }
        };
        f32x4 acc[3][2][NB][2];                                     // [gate R, Z, N][unit tile][row tile][16-row half]
        auto bias_set = [&](int set, f32x4 (&b)[2]) {               // b[T][i] = bias of unit 8 q + 4 T + i
            const char* bp = smem + (bias_off + set * 128 + ((opq(lane16) >> 8) << 5));
            const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 16);
            b[0] = f32x4{b0.x, b0.y, b0.z, b0.w};
            b[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
        };
        {
            f32x4 b0[2], b1[2];
            bias_set(0, b0);
            bias_set(1, b1);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { acc[0][T][bt][h] = b0[T]; acc[1][T][bt][h] = b1[T]; }
        }
        // the double pair's activation blobs (B operands of the scaled instruction): lanes 0-31 the even consumption's, lanes 32-63 the odd one's
        u32x6_t xc[2][NB];
        // E8M0 scale of the x blobs: x_hi * 4 (lanes q & 1 = 0) | x_lo * 2^14
        auto sb_x = [&]() -> uint32_t { return (opq(lane16) & 0x100) ? (uint32_t)kMxScaleLo : (uint32_t)kMxScaleHi; };
        auto rd_xh = [&](uint4 (&xh)[2][NB], int xs) {              // B operands of the main product: this lane's 16 bytes of the pair's hi fragments
            const int a = xs + f_lxs();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) xh[h][bt] = *reinterpret_cast<const uint4*>(smem + a + ((bt * 2) << 10) + h * 256);
        };
        auto rd_xh_half = [&](uint4 (&xh)[NB], int xs, int h) {     // ... of one 16-row half
            const int a = xs + f_lxs() + h * 256;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) xh[bt] = *reinterpret_cast<const uint4*>(smem + a + ((bt * 2) << 10));
        };
        // The blob of consumption c (its ring slot at xs), for the lanes that take it.  Odd c: lanes 32-63 only, lanes 0-31 keep the even
        // consumption's; even c: every lane (the upper lanes' registers are dead behind the previous double pair's correction products: no
        // branch).  Behind the exec-masked reads of an odd c the compiler's LDS counter is a lower bound (the join of the skip branch), so NO
        // LDS read may be in flight across them that is awaited later - it would be waited for together with all twelve blob reads: read at
        // the pair's start, in front of the main products, they cost 10 % of the kernel (profiles/r06_d).  They are issued behind the pair's
        // first row half, when nothing is pending, and the second half's operands behind them.
        auto rd_xc = [&](int xs, int parity) {
            if constexpr (kMx16Diag & 2) return;
            auto body = [&]() {
                const int a0 = xs + f_lpos();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int a = opq(a0 + h * 256);        // (one opaque base per row half keeps the two halves' 8-byte reads apart)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
                        xc[h][bt] = blob6(*reinterpret_cast<const uint4*>(smem + a + ((bt * 2 + 1) << 10)), rd8(smem + a + (((NB + bt) * 2 + 1) << 10)));
                }
            };
            if (parity == 0) body();
            else if ((opq(lane16) >> 9) == 1) body();
        };

        // ---------------- phase A: R, Z += W_i{r,z} x_t, pairs 0..15, software-pipelined across the pairs' barriers ---------------------
        // What barrier P publishes is pair P + 1's ring slot; what it retires is pair P's.  Pair P's operands are read from its slot in
        // front of barrier P (both row halves' hi fragments and the blob end up in registers), so HALF of its main products can be issued
        // behind that barrier - and hide the LDS latency of the next pair's first operands, which can only be requested there.  Iteration i =
        // what lies between barriers i - 1 and i (MFMAs: 12 + (24) + 12):
        //     odd i: rd blob of pair i for lanes 32-63           (LDS, 12 exec-masked reads: first, while nothing is pending - see rd_xc)
        //     rd hi fragments, row half 0, of pair i             (LDS, 3 reads)
        //     main products of pair i - 1, row half 1            (12; the operands were read in front of the barrier)
        //     rd hi fragments, row half 1, of pair i             (LDS, 3 reads: into the registers just consumed)
        //     weight requests: hi of pair i + 2 into the slot just consumed (pair i - 1's: slot (i - 1) % 3), even i: the blobs of double pair i / 2
        //     ring refill of pair i - 1's slot                   (behind the weight requests: a request behind it returns one HBM round trip later)
        //     even i: the correction products of double pair (i - 2, i - 1)  (24), then rd blob of pair i, every lane (LDS, 12 reads)
        //     main products of pair i, row half 0                (12)
        //     counted wait, barrier i                            (the compiler drains the LDS queue in front of a barrier: every read above is old by then)
        // The first form of this kernel (profiles/r06_b: every read at the pair's start, every request at its end) needed 2560 cycles per pair
        // against the 32-wide kernel's 1770: all eight waves read 84 KiB of LDS in one burst behind the barrier while the matrix pipe idled.
        // Pair 15's ring refill is deferred to the end of phase B (issued here it would sit in front of phase B's weight requests).
        int slot_a15 = 0, slot_prev = slot;
        uint4 xh[2][NB];
        // main products of pair P, row half H, fragment by fragment (f = 2 T + g: NB products each); `after(f)` runs behind fragment f's last use
        auto main_a = [&](auto pc, auto hc, auto&& after) {
            constexpr int P = decltype(pc)::value, H = decltype(hc)::value, WS = P % kMx16SA;
            static_for<0, 4>([&](auto FC) {
                constexpr int F = decltype(FC)::value, T = F >> 1, G = F & 1;
                if constexpr (!((kMx16Diag & 64) && P > 0)) {
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
                        acc[G][T][bt][H] = P == 0 ? mfma32k_first(wh[WS][F], xh[H][bt], acc[G][T][bt][H]) : mfma32k(wh[WS][F], xh[H][bt], acc[G][T][bt][H]);
                }
                CCSM_FENCE;
                after(FC);
                CCSM_FENCE;
            });
        };
        // correction products of double pair D (pairs 2 D, 2 D + 1) in six groups of four (one B operand each); `after(j)` runs behind group j
        auto corr_a = [&](auto dc, auto&& after) {
            constexpr int BS = decltype(dc)::value & 1;
            const uint32_t sbv = sb_x();
            static_for<0, 6>([&](auto JC) {
                constexpr int J = decltype(JC)::value, h = J / NB >= 2 ? 1 : J / NB, bt = J % NB;
                if constexpr (J < 2 * NB && !(kMx16Diag & 1)) {
                    const u32x6_t xb = xc[h][bt];
                    acc[0][0][bt][h] = mfma_corr16<0>(wb[BS][0], wsa[BS], xb, sbv, acc[0][0][bt][h]);
                    acc[1][0][bt][h] = mfma_corr16<1>(wb[BS][1], wsa[BS], xb, sbv, acc[1][0][bt][h]);
                    acc[0][1][bt][h] = mfma_corr16<2>(wb[BS][2], wsa[BS], xb, sbv, acc[0][1][bt][h]);
                    acc[1][1][bt][h] = mfma_corr16<3>(wb[BS][3], wsa[BS], xb, sbv, acc[1][1][bt][h]);
                }
                CCSM_FENCE;
                after(JC);
                CCSM_FENCE;
            });
        };
        static_for<0, NPAIR + 1>([&](auto IC) {
            constexpr int I = decltype(IC)::value;
            constexpr bool REFILL = I >= 1 && I < NPAIR, BLOB = mx16_has_blob(I);
            const int xs = X_OFF + slot * SLOT_BYTES;               // pair I's ring slot
            if constexpr (I < NPAIR && (I & 1)) rd_xc(xs, 1);       // (exec-masked: nothing may be pending in front of it - see rd_xc; the barrier drained the LDS queue)
            if constexpr (I < NPAIR) rd_xh_half(xh[0], xs, 0);
            CCSM_FENCE;
            if constexpr (I >= 1 && I < NPAIR) pstamp(I, 0);
            if constexpr (I >= 1) {
                // pair I - 1's second half; behind fragment f's last use its registers take fragment f of pair I + 2 (mx16_req_a(I, f) requests)
                main_a(std::integral_constant<int, I - 1>{}, std::integral_constant<int, 1>{}, [&](auto FC) {
                    constexpr int F = decltype(FC)::value;
                    if constexpr (I + 2 < NPAIR) wh[(I + 2) % kMx16SA][F] = w_at(((I + 2) >> 1) * PA + ((((I + 2) & 1) * 4 + F) << 10));
                    else if constexpr (I >= NPAIR - 1) {            // phase B's set I - 15: fragments 0, 1 | 2 | 3, 4 | 5
                        constexpr int U = I - (NPAIR - 1), K0 = (F >> 1) * 3 + (F & 1) * 2;
                        w[U][K0] = w_at(OFF_B + ((U * 6 + K0) << 10));
                        if constexpr (!(F & 1)) w[U][K0 + 1] = w_at(OFF_B + ((U * 6 + K0 + 1) << 10));
                    }
                });
                if constexpr (I < NPAIR) rd_xh_half(xh[1], xs, 1);
                CCSM_FENCE;
                if constexpr (I < NPAIR) pstamp(I, 1);
                if constexpr (REFILL) dma_ahead(slot_prev, s, I - 1, 0); else if constexpr (I == NPAIR) slot_a15 = slot_prev;
                CCSM_FENCE;
                if constexpr (I < NPAIR) pstamp(I, 2);
                if constexpr (!(I & 1)) {
                    // the double pair's correction products; behind their groups the next double pair's blobs and the refill's second request
                    corr_a(std::integral_constant<int, ((I - 1) >> 1)>{}, [&](auto JC) {
                        constexpr int J = decltype(JC)::value;
                        if constexpr (BLOB && J < 4) wb[(I >> 1) & 1][J] = w_at((I >> 1) * PA + ((8 + J) << 10));
                        if constexpr (BLOB && J == 4) wsa[(I >> 1) & 1] = ws_at((I >> 1) * PA + (12 << 10));
                        if constexpr (REFILL && J == (BLOB ? 5 : 0)) dma_ahead(slot_prev, s, I - 1, 1);
                    });
                }
                CCSM_FENCE;
                if constexpr (I < NPAIR) pstamp(I, 3);
            } else {
                rd_xh_half(xh[1], xs, 1);
            }
            if constexpr (I < NPAIR) {
                if constexpr (!(I & 1)) rd_xc(xs, 0);               // every lane: behind the correction products that read the registers
                CCSM_FENCE;
                main_a(IC, std::integral_constant<int, 0>{}, [&](auto FC) {
                    if constexpr (REFILL && (I & 1) && decltype(FC)::value == 1) dma_ahead(slot_prev, s, I - 1, 1);
                });
                CCSM_FENCE;
                if constexpr (I >= 1) pstamp(I, 4);
                xfer_wait(IC);
                if constexpr (I >= 1) pstamp(I, 5);
                if constexpr (!(kMx16Diag & 4)) __syncthreads();     // the next pair is in LDS; every wave has read this pair's operands
                if constexpr (I >= 1) pstamp(I, 6);
                slot_prev = slot;
                slot = slot == RS - 1 ? 0 : slot + 1;
            }
        });

        stamp(1);
        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn): twelve uses of six fragments per step ---------
        {
            f32x4 b3[2];
            bias_set(3, b3);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[2][T][bt][h] = b3[T];
        }
        static_for<0, 12>([&](auto UC) {
            constexpr int U = decltype(UC)::value;
            constexpr int D = U / 3, KIND = U % 3, SL = U & 1;
            if constexpr (KIND < 2) {                               // main products of pair 2 D + KIND
                constexpr int P = 2 * D + KIND;
                uint4 xh[2][NB];                                    // both row halves requested up front: the second is in flight under the first's 18 MFMAs
                rd_xh(xh, ((2 * P * NB) * 2) << 10);
                CCSM_FENCE;
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                            for (int g = 0; g < 3; ++g)
                                acc[g][T][bt][H] = (U == 0 && g == 2) ? mfma32k_first(w[SL][3 * T + g], xh[H][bt], acc[g][T][bt][H]) : mfma32k(w[SL][3 * T + g], xh[H][bt], acc[g][T][bt][H]);
                    CCSM_FENCE;
                });
            } else {                                                // correction products of the double pair D
                // E8M0 scale of the state's blobs; the initial states were packed kMxH0Div coarser (first step)
                const uint32_t sbv = sb_x() + (uint32_t)(s == 0 ? kMxScaleHi0 - kMxScaleHi : 0);
                u32x6_t xb[2][NB];
                {
                    const int a = f_lbs();
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt)
                            xb[h][bt] = blob6(*reinterpret_cast<const uint4*>(smem + a + ((((4 * D) * NB + bt) * 2 + 1) << 10) + h * 256),
                                              rd8(smem + a + ((((4 * D + 1) * NB + bt) * 2 + 1) << 10) + h * 256));
                }
                CCSM_FENCE;
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) {
                            acc[0][T][bt][H] = mfma_corr16<0>(w[SL][3 * T + 0], wsc[T], xb[H][bt], sbv, acc[0][T][bt][H]);
                            acc[1][T][bt][H] = mfma_corr16<1>(w[SL][3 * T + 1], wsc[T], xb[H][bt], sbv, acc[1][T][bt][H]);
                            acc[2][T][bt][H] = mfma_corr16<2>(w[SL][3 * T + 2], wsc[T], xb[H][bt], sbv, acc[2][T][bt][H]);
                        }
                    CCSM_FENCE;
                });
            }
            if constexpr (U + 2 < 12) ld_set(std::integral_constant<int, U + 2>{});      // the slot just used takes the set of two uses ahead
            else if constexpr (U == 10) { ldC_hi(std::integral_constant<int, 0>{}); ldC_hi(std::integral_constant<int, 1>{}); }
            CCSM_FENCE;
        });
        dma_ahead(slot_a15, s, NPAIR - 1);                          // the deferred ring refill (both requests): phase-C position RS - 1
        // r = sigmoid(R) ; N = b_in + r * N
        mfma_drain();
        mfma_drained<NB>(acc[0]);
        mfma_drained<NB>(acc[1]);
        mfma_drained<NB>(acc[2]);
        {
            f32x4 b2[2];
            bias_set(2, b2);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2][T][bt][h][i] = b2[T][i] + ((kMx16Diag & 128) ? acc[0][T][bt][h][i] : sigmoid_f(acc[0][T][bt][h][i])) * acc[2][T][bt][h][i];
                        asm volatile("" : "+v"(acc[2][T][bt][h]));  // pins the evaluation HERE: sunk to phase C's first products, R stayed alive across the
                                                                    // requests below and they were spilled, each behind an s_waitcnt vmcnt(0)
                    }
        }
        // phase C's positions 2, 3 and the blobs of its first double pair, once R is dead (the accumulators drop from 144 to 96 registers)
        CCSM_FENCE;
        ldC_hi(std::integral_constant<int, 2>{});
        ldC_hi(std::integral_constant<int, 3>{});
        ldC_blob(std::integral_constant<int, 0>{});
        CCSM_FENCE;
        auto zwork = [&](int bt) {                                  // z = sigmoid(Z) in place, inside phase C (vector ALU otherwise idle)
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = (kMx16Diag & 128) ? acc[1][T][bt][h][i] * 0.5f : sigmoid_f(acc[1][T][bt][h][i]);
                        asm volatile("" : "+v"(v));                 // pins the evaluation HERE (the compiler otherwise sinks it to the tail)
                        acc[1][T][bt][h][i] = v;
                    }
        };

        stamp(2);
        // ---------------- phase C: N += W_in x_t, positions 0..15 (consumptions 16..31, zig-zag), pipelined like phase A (MFMAs per iteration:
        // 6 + (12) + 6); position pp's hi fragments live in slot pp & 3 and are refilled with position pp + 4's behind their last use; double pair
        // DC's blobs in slot DC & 1; the last four iterations request the next step's phase-A pairs 0 and 1 instead ---------------------------
        auto main_c = [&](auto cc_, auto hc, auto&& after) {        // main products of position CC, row half H: fragment T (NB products), then after(T)
            constexpr int CC = decltype(cc_)::value, H = decltype(hc)::value, WS = CC & 3;
            static_for<0, 2>([&](auto TC) {
                constexpr int T = decltype(TC)::value;
                if constexpr (!((kMx16Diag & 64) && CC > 0)) {
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
                        acc[2][T][bt][H] = CC == 0 ? mfma32k_first(wch[WS][T], xh[H][bt], acc[2][T][bt][H]) : mfma32k(wch[WS][T], xh[H][bt], acc[2][T][bt][H]);
                }
                CCSM_FENCE;
                after(TC);
                CCSM_FENCE;
            });
        };
        auto corr_c = [&](auto dc, auto&& after) {                  // six groups of two
            constexpr int BS = decltype(dc)::value & 1;
            const uint32_t sbv = sb_x();
            const u32x6_t a0 = blob6(wcb0[BS][0], wcb1[BS][0]), a1 = blob6(wcb0[BS][1], wcb1[BS][1]);
            static_for<0, 6>([&](auto JC) {
                constexpr int J = decltype(JC)::value, h = J / NB >= 2 ? 1 : J / NB, bt = J % NB;
                if constexpr (J < 2 * NB && !(kMx16Diag & 1)) {
                    const u32x6_t xb = xc[h][bt];
                    acc[2][0][bt][h] = mfma_corr16_a6<0>(a0, wcs[BS], xb, sbv, acc[2][0][bt][h]);
                    acc[2][1][bt][h] = mfma_corr16_a6<1>(a1, wcs[BS], xb, sbv, acc[2][1][bt][h]);
                }
                CCSM_FENCE;
                after(JC);
                CCSM_FENCE;
            });
        };
        slot_prev = slot;
        static_for<0, NPAIR + 1>([&](auto IC) {
            constexpr int I = decltype(IC)::value;
            constexpr bool BLOB = mx16_has_blob(I);
            const int xs = X_OFF + slot * SLOT_BYTES;               // position I's ring slot
            if constexpr (I < NPAIR && (I & 1)) rd_xc(xs, 1);
            if constexpr (I < NPAIR) rd_xh_half(xh[0], xs, 0);
            CCSM_FENCE;
            if constexpr (I >= 1) {
                // position I - 1's second half; behind fragment T's last use its registers take fragment T of position I + 3 (the last four
                // iterations: of the next step's phase-A pairs 0 and 1)
                main_c(std::integral_constant<int, I - 1>{}, std::integral_constant<int, 1>{}, [&](auto TC) {
                    constexpr int T = decltype(TC)::value;
                    if constexpr (I + 3 < NPAIR) wch[(I + 3) & 3][T] = w_at(OFF_C + ((I + 3) >> 1) * PC + ((((I + 3) & 1) * 2 + T) << 10));
                    else {
                        constexpr int p = (I + 3 - NPAIR) >> 1, F = 2 * ((I + 3 - NPAIR) & 1) + T;
                        wh[p % kMx16SA][F] = w_at((p >> 1) * PA + (((p & 1) * 4 + F) << 10));
                    }
                });
                if constexpr (I < NPAIR) rd_xh_half(xh[1], xs, 1);
                CCSM_FENCE;
                dma_ahead(slot_prev, s, NPAIR + I - 1, 0);
                CCSM_FENCE;
                if constexpr (!(I & 1)) {
                    corr_c(std::integral_constant<int, ((I - 1) >> 1)>{}, [&](auto JC) {
                        constexpr int J = decltype(JC)::value, DC = I >> 1;
                        if constexpr (BLOB && J == 0) wcb0[DC & 1][0] = w_at(OFF_C + DC * PC + (4 << 10));
                        if constexpr (BLOB && J == 1) wcb1[DC & 1][0] = w8_at(OFF_C + DC * PC + (6 << 10));
                        if constexpr (BLOB && J == 2) wcb0[DC & 1][1] = w_at(OFF_C + DC * PC + (5 << 10));
                        if constexpr (BLOB && J == 3) wcb1[DC & 1][1] = w8_at(OFF_C + DC * PC + (6 << 10) + 512);
                        if constexpr (BLOB && J == 4) wcs[DC & 1] = ws_at(OFF_C + DC * PC + (7 << 10));
                        if constexpr (J == (BLOB ? 5 : 0)) dma_ahead(slot_prev, s, NPAIR + I - 1, 1);
                    });
                }
                CCSM_FENCE;
                if constexpr (I - 1 == 1) zwork(0);
                if constexpr (I - 1 == 5 && NB > 1) zwork(1);
                if constexpr (I - 1 == 9 && NB > 2) zwork(2);
            } else {
                rd_xh_half(xh[1], xs, 1);
            }
            if constexpr (I < NPAIR) {
                if constexpr (!(I & 1)) rd_xc(xs, 0);
                CCSM_FENCE;
                main_c(IC, std::integral_constant<int, 0>{}, [&](auto TC) {
                    if constexpr (I >= 1 && (I & 1) && decltype(TC)::value == 0) dma_ahead(slot_prev, s, NPAIR + I - 1, 1);
                });
                CCSM_FENCE;
                xfer_wait(std::integral_constant<int, NPAIR + I>{});
                if constexpr (!(kMx16Diag & 4)) __syncthreads();
                slot_prev = slot;
                slot = slot == RS - 1 ? 0 : slot + 1;
            }
        });
        stamp(3);
        mfma_drain();
        mfma_drained<NB>(acc[2]);
        // ---------------- tail: n = tanh(N); h' = n + z (h_{t-1} - n); hi fragments, blobs and residuals for the next step and layer ----
        {
            const int a = f_lxs(), l16 = opq(lane16);
            const int lo_ = f_lpos() + ((l16 & 0x200) << 2);    // lxs inside the output's [kb][hl] fragments (one row tile per k-block)
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                char* p_hi = smem + own + a + bt * 2048;                                     // own 16 bytes of the hi fragments (half 0; + 256: half 1)
                char* p_c0 = smem + own + ((bt * 2 + 1) << 10) + l16;                         // blob bytes 0-15: corr fragment of the wave's first k-block
                char* p_c1 = smem + own + (((NB + bt) * 2 + 1) << 10) + l16;                  // bytes 16-23 | this lane's residuals of half 0
                char* p_lo = smem + LO_OFF + ((wave * NB + bt) * 64) * 8 + (l16 >> 1);        // residuals of half 1
                const uint2 l8[2] = {make_uint2(reinterpret_cast<const uint4*>(p_c1)->z, reinterpret_cast<const uint4*>(p_c1)->w),
                                     *reinterpret_cast<const uint2*>(p_lo)};
                float hn[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const half8 hv = as_half8(*reinterpret_cast<const uint4*>(p_hi + h * 256));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int lo4 = (int)(j < 4 ? l8[h].x : l8[h].y);
                        const float lo = (j & 3) == 0 ? __builtin_amdgcn_cvt_f32_fp8(lo4, 0) : (j & 3) == 1 ? __builtin_amdgcn_cvt_f32_fp8(lo4, 1)
                                       : (j & 3) == 2 ? __builtin_amdgcn_cvt_f32_fp8(lo4, 2) : __builtin_amdgcn_cvt_f32_fp8(lo4, 3);
                        const float hp = (float)hv[j] + lo * (1.0f / kMxLoScale);
                        const float nn = (kMx16Diag & 128) ? acc[2][j >> 2][bt][h][j & 3] * 0.001f : tanh_fold(acc[2][j >> 2][bt][h][j & 3]);
                        hn[h][j] = (hp - nn) * acc[1][j >> 2][bt][h][j & 3] + nn;
                    }
                }
                uint4 hi[2], c0;
                uint2 lo8[2], c1;
                mx16_pack<false>(hn, 0.25f, hi, lo8, c0, c1);
                const uint4 c1w = make_uint4(c1.x, c1.y, lo8[0].x, lo8[0].y);
                *reinterpret_cast<uint4*>(p_hi) = hi[0];
                *reinterpret_cast<uint4*>(p_hi + 256) = hi[1];
                *reinterpret_cast<uint4*>(p_c0) = c0;
                *reinterpret_cast<uint4*>(p_c1) = c1w;
                *reinterpret_cast<uint2*>(p_lo) = lo8[1];
                // streaming stores: the next reader is another kernel 0.5 GB later, keep the L2 for the weight stream
                char* o = reinterpret_cast<char*>(out + (((size_t)(tile0 + bt) * kSeqLen + t) * kKB12 + (dir * kKBH + 2 * wave)) * 2 * kFragU4);
                nt_store(hi[0], reinterpret_cast<uint4*>(o + (uint32_t)lo_));
                nt_store(hi[1], reinterpret_cast<uint4*>(o + (uint32_t)lo_ + 256));
                if constexpr (OUT_FP8) {
                    // For the attention pool (attn_fc_f8_kernel): ONE compact residual fragment per pair - lane (n, g) holds the 16 fp8 residuals
                    // (x 2^17) of k-block g of row n, byte j <-> k = kCorrPerm[j] (dwords: k 0-3 | 8-11 | 4-7 | 12-15).  Lane (n', q) holds
                    // k = 8 (q & 1) + 0..7 of k-block q >> 1 for rows n' (half 0) and 16 + n' (half 1), and IS fragment lane (16 (q & 1) + n',
                    // q >> 1): the even lane groups keep their half 0 and take the odd neighbour's, the odd ones their half 1 - one
                    // v_permlane16_swap per dword.
                    uint32_t r8[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            float lf[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = hn[h][4 * u + e];
                                lf[e] = __builtin_amdgcn_fmed3f(v - (float)(_Float16)v, -kF8Clamp / kCorrActLo, kF8Clamp / kCorrActLo);
                            }
                            r8[h][u] = cvt4_fp8_l(hi[0].x, lf[0], lf[1], lf[2], lf[3]);
                        }
                    swap16(r8[0][0], r8[1][0]);     // even groups: (own half 0, odd's half 0); odd groups: (even's half 1, own half 1)
                    swap16(r8[0][1], r8[1][1]);
                    nt_store(make_uint4(r8[0][0], r8[1][0], r8[0][1], r8[1][1]), reinterpret_cast<uint4*>(o + 1024 + (uint32_t)l16));
                } else {
                    nt_store(c0, reinterpret_cast<uint4*>(o + 1024 + (uint32_t)l16));
                    nt_store(c1w, reinterpret_cast<uint4*>(o + 3072 + (uint32_t)l16));   // (bytes 8-15 are spare in HBM)
                }
            }
        }
        CCSM_FENCE;
        // the rest of what the next step's first pairs need (not live across the tail): hi of pair 2, the blobs of double pair 0
        ldA_hi2(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
        ldA_hi2(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
        ldA_blob(std::integral_constant<int, 0>{});
        CCSM_FENCE;
        stamp(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
}
#undef CCSM_FENCE

}  // namespace ccsm
