// Shared host/device layout constants for libccsm (gfx950).
//
// Everything the MFMA kernels consume lives in HBM as *fragments*: one fragment = the operand of one
// v_mfma_f32_32x32x16_f16 for one wavefront = 64 lanes x 8 halfs (16 B per lane, 1 KiB), lane-linear, so that a
// wave loads it with one coalesced dwordx4 per lane (global) or one conflict-free ds_read_b128 (LDS).
//
//   lane l = n + 32*g  (n = 0..31, g = 0..1) holds M[n][16*kb + 8*g + j], j = 0..7
//
// where for a weight (A) fragment n indexes 32 consecutive output units of one gate and for an activation (B)
// fragment n indexes 32 consecutive batch rows; k runs over the contraction dimension in blocks (kb) of 16.
// Every fp32 value v is carried as an fp16 pair hi = fp16(v), lo = fp16(v - hi) (two fragments, "hl" = 0/1).
#pragma once
#include <stdint.h>

namespace ccsm {

constexpr int kSeqLen = 21;        // L: k-mer length (reference --seq_len 21)
constexpr int kHidden = 256;       // H: --hid_rnn 256
constexpr int kLayers = 3;         // --layer_rnn 3
constexpr int kGates = 3;          // r, z, n (torch.nn.GRU row order)
constexpr int kClasses = 2;
constexpr int kEmbed = 8;          // NEMBED_BASE
constexpr int kVocab = 5;          // N_VOCAB
constexpr int kFeat0 = 11;         // 8 embedding dims + ipd + pw + npass
constexpr int kWaves = 8;          // waves per workgroup; wave w owns hidden units [32w, 32w+32)
constexpr int kUnitTile = 32;
constexpr int kKB0 = 1;            // k-blocks of the layer-0 input (11 padded to 16)
constexpr int kKBH = kHidden / 16; // 16 k-blocks of the recurrent input
constexpr int kKB12 = 2 * kHidden / 16;  // 32 k-blocks of the layer-1/2 input (fwd|bwd concat)
constexpr int kFragU4 = 64;        // uint4 per fragment
constexpr int kAttHidden = 256;

// number of k-blocks streamed per step by one wave in layer `l`
inline constexpr int layer_kx(int l) { return l == 0 ? kKB0 : kKB12; }
// uint4 count of one (layer, dir, wave) weight stream
inline constexpr size_t wstream_u4_per_wave(int l) { return (size_t)(layer_kx(l) + kKBH) * kGates * 2 * kFragU4; }

}  // namespace ccsm
