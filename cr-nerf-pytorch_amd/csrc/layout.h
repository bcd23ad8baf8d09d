// Packed-weight layout shared by the pack kernel, the MLP core and the host shim.
//
// Reference weight layout being re-packed: NeRF_sigma.__init__, models/nerf.py:116-154
// (24 nn.Linear tensors, weight [out,in] row-major, bias [out]).
//
// One packed model = [ consts | weight stream ].
//
//   consts  : biases + the sigma head, compact fp32 (CONST_FLOATS floats, padded to CONST_BYTES)
//   stream  : a flat sequence of 1 KiB "fragments".  One fragment is exactly what one wavefront
//             reads with a single ds_read_b128 (64 lanes x 4 floats) and feeds to four
//             v_mfma_f32_32x32x2_f32 as the A operand:
//                 frag(layer, group v, tile t)[lane = 32*kk + i][j] = W[32*t + i][col(8*v + 4*kk + j)]
//             i.e. output-feature row 32t+i, and the four k-values this lane supplies for the
//             four MFMAs of k-group v.  Fragments are ordered group-major, tile-minor inside a
//             layer, layers in execution order, and cut into 16-fragment (16 KiB) "stages" that
//             the kernels stream through an LDS ring with global_load_lds.
//
// "col(k)" maps the kernel's k order onto the reference's input-column order; it is the identity
// for hidden activations and the pair-interleaved order of posenc_slots.h for the two positional
// embeddings (so that one lane computes sin AND cos of the same argument with one sincosf).
#pragma once
#include <stdint.h>

namespace crnerf {

constexpr int W_HIDDEN = 256;
constexpr int XYZ_FREQS = 15;             // N_emb_xyz, opt.py:46
constexpr int DIR_FREQS = 4;              // N_emb_dir, opt.py:48
constexpr int XYZ_DIM = 6 * XYZ_FREQS + 3;  // 93
constexpr int DIR_DIM = 6 * DIR_FREQS + 3;  // 27
constexpr int XYZ_PAD = 96;
constexpr int DIR_PAD = 32;
constexpr int IN_DIM = XYZ_DIM + DIR_DIM;   // 120
constexpr int FEAT_DIM = 64;                // nerf_out_dim, opt.py:93
constexpr int OUT_DIM = FEAT_DIM + 1;       // 65

constexpr int FRAG_FLOATS = 256;
constexpr int FRAG_BYTES = 1024;
constexpr int STAGE_FRAGS = 16;
constexpr int STAGE_BYTES = STAGE_FRAGS * FRAG_BYTES;  // 16 KiB

// k-groups (8 k-values each) and output tiles (32 rows each) per layer
constexpr int G_XYZ = XYZ_PAD / 8;   // 12
constexpr int G_HID = W_HIDDEN / 8;  // 32
constexpr int G_DIR = DIR_PAD / 8;   // 4
constexpr int G_HALF = 128 / 8;      // 16

// fragments per layer, in execution order
constexpr int FR_L1 = G_XYZ * 8;            //  96  xyz_encoding_1      93(96)->256
constexpr int FR_HID = G_HID * 8;           // 256  xyz_encoding_{2,3,4,6,7,8}, xyz_encoding_final
constexpr int FR_L5 = (G_XYZ + G_HID) * 8;  // 352  xyz_encoding_5      [xyz(96), h(256)]->256
constexpr int FR_DIR = (G_HID + G_DIR) * 4; // 144  dir_encoding        [final(256), dir(32)]->128
constexpr int FR_RGB = G_HALF * 2;          //  32  static_rgb          128->64

constexpr int OFF_L1 = 0;
constexpr int OFF_L2 = OFF_L1 + FR_L1;      // L2,L3,L4 contiguous
constexpr int OFF_L5 = OFF_L2 + 3 * FR_HID;
constexpr int OFF_L6 = OFF_L5 + FR_L5;      // L6,L7,L8 contiguous
constexpr int OFF_FIN = OFF_L6 + 3 * FR_HID;
constexpr int OFF_DIR = OFF_FIN + FR_HID;
constexpr int OFF_RGB = OFF_DIR + FR_DIR;
constexpr int STREAM_FRAGS = OFF_RGB + FR_RGB;               // 2416
constexpr int STAGES_PER_PASS = STREAM_FRAGS / STAGE_FRAGS;  // 151
static_assert(STREAM_FRAGS % STAGE_FRAGS == 0, "stream must be whole stages");
static_assert(FR_L1 % STAGE_FRAGS == 0 && FR_L5 % STAGE_FRAGS == 0 && FR_DIR % STAGE_FRAGS == 0 &&
              FR_RGB % STAGE_FRAGS == 0, "layers must be whole stages");

// consts block (float offsets)
constexpr int C_BIAS = 0;                    // 8 x 256: xyz_encoding_{1..8} biases
constexpr int C_BFIN = 8 * W_HIDDEN;         // 256: xyz_encoding_final bias
constexpr int C_WSIG = C_BFIN + W_HIDDEN;    // 256: static_sigma weight row
constexpr int C_BSIG = C_WSIG + W_HIDDEN;    // 1 (+3 pad): static_sigma bias
constexpr int C_BDIR = C_BSIG + 4;           // 128: dir_encoding bias
constexpr int C_BRGB = C_BDIR + 128;         // 64: static_rgb bias
constexpr int CONST_FLOATS = C_BRGB + 64;    // 2756
constexpr int CONST_BYTES = 11264;           // padded to 11 KiB (multiple of 1 KiB)
static_assert(CONST_FLOATS * 4 <= CONST_BYTES, "consts overflow");

constexpr size_t PACKED_BYTES = (size_t)CONST_BYTES + (size_t)STREAM_FRAGS * FRAG_BYTES;  // 2,485,248

// Positional-embedding slot order.  An embedding with F frequencies occupies 8*ceil((3F+2)/4)
// padded k-slots.  Slot k = 8q + 4h + 2p + sc belongs to "argument" a = 4q + 2h + p:
//   a <  3F     : (sin, cos)[sc] of 2^(a/3) * x[a%3]   -> reference column 3 + 6(a/3) + 3sc + a%3
//   a == 3F     : (x, y)[sc]                           -> reference column sc
//   a == 3F + 1 : (z, 0)[sc]                           -> reference column 2 / pad
//   otherwise   : zero pad
// Reference column order: PosEmbedding.forward, models/nerf.py:17-30.
__host__ __device__ inline int posenc_slot_to_col(int k, int F) {
  const int q = k >> 3, h = (k >> 2) & 1, p = (k >> 1) & 1, sc = k & 1;
  const int a = 4 * q + 2 * h + p;
  if (a < 3 * F) return 3 + 6 * (a / 3) + 3 * sc + (a % 3);
  if (a == 3 * F) return sc;
  if (a == 3 * F + 1) return sc == 0 ? 2 : -1;
  return -1;
}

// ---- "v16" fragment order for the v_mfma_f32_16x16x4_f32 core (mlp_core16.h): same layers, same
// fragment counts and stage cuts, but a fragment is
//     frag(layer, k-group u (16 k-values), tile T (16 rows))[lane = 16*kq + i][r] = W[16T + i][col(16u + 4kq + r)]
// and the embedding slot order pairs (sin, cos) inside a 4-lane-group: slot k = 16v + 4g + 2p + sc
// belongs to argument a = 8v + 2g + p (same argument -> column rule as posenc_slot_to_col).
__host__ __device__ inline int posenc_slot_to_col16(int k, int F) {
  const int v = k >> 4, g = (k >> 2) & 3, p = (k >> 1) & 1, sc = k & 1;
  const int a = 8 * v + 2 * g + p;
  if (a < 3 * F) return 3 + 6 * (a / 3) + 3 * sc + (a % 3);
  if (a == 3 * F) return sc;
  if (a == 3 * F + 1) return sc == 0 ? 2 : -1;
  return -1;
}

// ---- transposed ("WT") stream for the backward-data pass (mlp_backward16.hip), v16 fragment shape:
//     fragT(layer, k-group u (16 OUTPUT features = the reduction), tile T (16 INPUT features))[lane = 16*kq + i][r]
//         = W[16u + 4kq + r][in_off + 16T + i]
// layers in backward order; only the hidden part of each input is needed (embeddings are not trainable).
constexpr int FT_RGB = 4 * 8;      //  32: static_rgb^T      64 -> 128
constexpr int FT_DIR = 8 * 16;     // 128: dir_encoding^T   128 -> 256 (columns 0..255 of the [128,283] weight)
constexpr int FT_HID = 16 * 16;    // 256: xyz_encoding_final^T and xyz_encoding_{8..2}^T (layer 5: columns 93..348)
constexpr int OFFT_RGB = 0;
constexpr int OFFT_DIR = OFFT_RGB + FT_RGB;
constexpr int OFFT_FIN = OFFT_DIR + FT_DIR;
constexpr int OFFT_L8 = OFFT_FIN + FT_HID;          // L8, L7, ..., L2 contiguous
constexpr int STREAMT_FRAGS = OFFT_L8 + 7 * FT_HID; // 2208
constexpr int STAGEST_PER_PASS = STREAMT_FRAGS / STAGE_FRAGS;  // 138
static_assert(STREAMT_FRAGS % STAGE_FRAGS == 0, "transposed stream must be whole stages");
constexpr size_t PACKEDT_BYTES = (size_t)CONST_BYTES + (size_t)STREAMT_FRAGS * FRAG_BYTES;

// ---- bf16 stream for the v_mfma_f32_32x32x16_bf16 core (mlp_core_bf16.h).  A fragment is still 1 KiB = one
// ds_read_b128 per wavefront, now the A operand of ONE MFMA (32 output rows x 16 k-values, bf16):
//     fragB(layer, tile T (32 rows), k-step s (16 k-values))[lane = 32*hh + i][e] = bf16(W[32T + i][col(kB(s, hh, e))])
// ordered layer-major, tile-major, k-step-minor (a tile's accumulators complete after its k-steps, so its
// epilogue overlaps the next tile's MFMAs and only one tile of accumulators is live).
//   * k-steps fed by hidden activations: kB = feature 32(s/2) + 16(s%2) + 4hh + (e&3) + 8(e>>2) -- exactly what
//     lane-half hh holds in registers 8(s%2)..8(s%2)+7 of source tile s/2 after the 32x32 C/D layout.
//   * k-steps fed by an embedding: slot 16s + 8hh + e, e = 2p + sc, belongs to argument a = 8s + 4hh + p
//     (same argument -> column rule as posenc_slot_to_col).
// consts block: the fp32 one (biases and the sigma head stay fp32).  The stream is padded to whole stages.
constexpr int KS_XYZ = XYZ_PAD / 16;   // 6 k-steps
constexpr int KS_HID = W_HIDDEN / 16;  // 16
constexpr int KS_DIR = DIR_PAD / 16;   // 2
constexpr int KS_HALF = 128 / 16;      // 8
constexpr int FB_L1 = 8 * KS_XYZ;              //  48
constexpr int FB_HID = 8 * KS_HID;             // 128
constexpr int FB_L5 = 8 * (KS_XYZ + KS_HID);   // 176
constexpr int FB_DIR = 4 * (KS_HID + KS_DIR);  //  72
constexpr int FB_RGB = 2 * KS_HALF;            //  16
constexpr int OFFB_L1 = 0;
constexpr int OFFB_L2 = OFFB_L1 + FB_L1;       // L2, L3, L4 contiguous
constexpr int OFFB_L5 = OFFB_L2 + 3 * FB_HID;
constexpr int OFFB_L6 = OFFB_L5 + FB_L5;       // L6, L7, L8 contiguous
constexpr int OFFB_FIN = OFFB_L6 + 3 * FB_HID;
constexpr int OFFB_DIR = OFFB_FIN + FB_HID;
constexpr int OFFB_RGB = OFFB_DIR + FB_DIR;
constexpr int STREAMB_USED = OFFB_RGB + FB_RGB;                                             // 1208
constexpr int STAGESB_PER_PASS = (STREAMB_USED + STAGE_FRAGS - 1) / STAGE_FRAGS;            // 76
constexpr int STREAMB_FRAGS = STAGESB_PER_PASS * STAGE_FRAGS;                               // 1216 (8 zero fragments)
constexpr size_t PACKEDB_BYTES = (size_t)CONST_BYTES + (size_t)STREAMB_FRAGS * FRAG_BYTES;  // 1,256,448

__host__ __device__ inline int posenc_slot_to_col_b(int k, int F) {
  const int s = k >> 4, hh = (k >> 3) & 1, p = (k >> 1) & 3, sc = k & 1;
  const int a = 8 * s + 4 * hh + p;
  if (a < 3 * F) return 3 + 6 * (a / 3) + 3 * sc + (a % 3);
  if (a == 3 * F) return sc;
  if (a == 3 * F + 1) return sc == 0 ? 2 : -1;
  return -1;
}

// ---- "x3" stream for the fp32-accurate core on the bf16 matrix cores (mlp_core_x3.h): every weight is THREE bf16 pieces,
// w = w1 + w2 + w3 with w1 = bf16(w), w2 = bf16(w - w1), w3 = bf16(w - w1 - w2); a fragment is the A operand of one
// v_mfma_f32_32x32x16_bf16 as in fragB, and the stream is ordered layer-major, K-STEP-major, tile-minor, piece-minor -- the core keeps
// the accumulators of all output tiles of a layer and walks the contraction once, splitting each fp32 activation once:
//     fragX(layer, k-step s, tile T, piece w)[lane = 32*hh + i][e] = piece_w(W[32T + i][col(16s + 8(e>>2) + 4hh + (e&3))])
// col(k): k is a hidden feature index (exactly the feature lane-half hh holds in registers 8(s%2) + e of source tile s/2 in the 32x32
// C/D layout) or, for the two embeddings, a padded slot in the fp32 order of posenc_slot_to_col -- the registers posenc_regs fills.
// dir_encoding (216 fragments) is padded to 240: whole 16-fragment stages AND whole 6-fragment turns of the core's read-ahead queue, whose
// phase must be the same at every layer entry; consts block: the fp32 one.
constexpr int FX_L1 = KS_XYZ * 8 * 3;                //  144
constexpr int FX_HID = KS_HID * 8 * 3;               //  384
constexpr int FX_L5 = (KS_XYZ + KS_HID) * 8 * 3;     //  528
constexpr int FX_DIR_USED = (KS_HID + KS_DIR) * 4 * 3;   // 216
constexpr int FX_DIR = (FX_DIR_USED + 47) / 48 * 48;   // 240
constexpr int FX_RGB = KS_HALF * 2 * 3;              //   48
constexpr int OFFX_L1 = 0;
constexpr int OFFX_L2 = OFFX_L1 + FX_L1;             // L2, L3, L4 contiguous
constexpr int OFFX_L5 = OFFX_L2 + 3 * FX_HID;
constexpr int OFFX_L6 = OFFX_L5 + FX_L5;             // L6, L7, L8 contiguous
constexpr int OFFX_FIN = OFFX_L6 + 3 * FX_HID;
constexpr int OFFX_DIR = OFFX_FIN + FX_HID;
constexpr int OFFX_RGB = OFFX_DIR + FX_DIR;
constexpr int STREAMX_FRAGS = OFFX_RGB + FX_RGB;                 // 3648
constexpr int STAGESX_PER_PASS = STREAMX_FRAGS / STAGE_FRAGS;    // 228
static_assert(STREAMX_FRAGS % STAGE_FRAGS == 0 && FX_L1 % STAGE_FRAGS == 0 && FX_HID % STAGE_FRAGS == 0 && FX_L5 % STAGE_FRAGS == 0 &&
              FX_RGB % STAGE_FRAGS == 0 && FX_DIR % STAGE_FRAGS == 0, "x3 layers must be whole stages");
static_assert(FX_L1 % 6 == 0 && FX_HID % 6 == 0 && FX_L5 % 6 == 0 && FX_DIR % 6 == 0 && FX_RGB % 6 == 0, "x3 layers must be whole queue turns");
constexpr size_t PACKEDX_BYTES = (size_t)CONST_BYTES + (size_t)STREAMX_FRAGS * FRAG_BYTES;   // 3,746,816

// ---- "fragH": the x3 stream with TWO fp16 pieces per weight ("h2" core: mlp_core_x3.h built with CRNERF_X_NP = 2).  An fp32 value is
// h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1) to 2^-24 of it (11 + 11 mantissa bits + the sign of h2: one fp32 rounding), as long as h2 stays
// a normal fp16 number; below that its absolute error is <= 2^-25 (fp16 subnormals are honoured by the matrix cores, tools/ubench/
// mfma_f16_denorm.hip).  Weights are therefore stored SCALED by H2_WSCALE = 2^8 (|w| >= 2^-10 keeps full precision, |w| < 255 stays in range;
// the core scales the bias up and the layer output down, both exact): piece 0 = fp16(256 w), piece 1 = fp16(256 w - piece 0).  Same
// fragment geometry and order as fragX with two pieces per (k-step, tile); dir_encoding needs no padding (144 = 9 stages = 36 queue turns).
constexpr float H2_WSCALE = 256.0f;
// the range flag of an h2 / h2t pack: one word of the pack's own consts block behind CONST_FLOATS (zeroed by pack_consts_kernel), set by the stream
// packers when a scaled weight is not a finite fp16 number.  The host reads it in crnerf_pack_mlp_weights_h2 / crnerf_pack_h2_status; the h2 kernels
// read it from the LDS copy of the consts: a refused pack poisons every ray (the f32x3 repair then renders all of them) / skips the h2 data gradient
// (the f32x3 one runs instead) -- precision "auto" without a host round trip.
constexpr int H2_FLAG_WORD = CONST_FLOATS;
static_assert((H2_FLAG_WORD + 1) * 4 <= CONST_BYTES, "no spare word in the consts block");
constexpr int FH_L1 = KS_XYZ * 8 * 2;                //   96
constexpr int FH_HID = KS_HID * 8 * 2;               //  256
constexpr int FH_L5 = (KS_XYZ + KS_HID) * 8 * 2;     //  352
constexpr int FH_DIR = (KS_HID + KS_DIR) * 4 * 2;    //  144
constexpr int FH_RGB = KS_HALF * 2 * 2;              //   32
constexpr int OFFH_L1 = 0;
constexpr int OFFH_L2 = OFFH_L1 + FH_L1;
constexpr int OFFH_L5 = OFFH_L2 + 3 * FH_HID;
constexpr int OFFH_L6 = OFFH_L5 + FH_L5;
constexpr int OFFH_FIN = OFFH_L6 + 3 * FH_HID;
constexpr int OFFH_DIR = OFFH_FIN + FH_HID;
constexpr int OFFH_RGB = OFFH_DIR + FH_DIR;
constexpr int STREAMH_FRAGS = OFFH_RGB + FH_RGB;                 // 2416
static_assert(STREAMH_FRAGS % STAGE_FRAGS == 0 && FH_L1 % STAGE_FRAGS == 0 && FH_HID % STAGE_FRAGS == 0 && FH_L5 % STAGE_FRAGS == 0 &&
              FH_RGB % STAGE_FRAGS == 0 && FH_DIR % STAGE_FRAGS == 0, "h2 layers must be whole stages (and with them whole 4-fragment queue turns)");
constexpr size_t PACKEDH_BYTES = (size_t)CONST_BYTES + (size_t)STREAMH_FRAGS * FRAG_BYTES;   // 2,485,248

// ---- transposed x3 stream for the backward-data pass on the x3 core (mlp_backward_x3.hip): fragX of M = (W restricted to the hidden inputs)^T,
//     fragXT(layer, k-step s, tile T, piece w)[lane = 32*hh + i][e] = piece_w(W[16s + 8(e>>2) + 4hh + (e&3)][in_off + 32T + i])
// (the contraction runs over the layer's OUTPUT features -- the delta registers, in the 32x32 C/D order -- the tiles over its inputs); layers in
// backward order, as the fp32 transposed stream (OFFT_*).
constexpr int FXT_RGB = (FEAT_DIM / 16) * 4 * 3;     //  48: static_rgb^T      64 -> 128
constexpr int FXT_DIR = (128 / 16) * 8 * 3;          // 192: dir_encoding^T   128 -> 256 (columns 0..255 of the [128,283] weight)
constexpr int FXT_HID = KS_HID * 8 * 3;              // 384: xyz_encoding_final^T and xyz_encoding_{8..2}^T (layer 5: columns 93..348)
constexpr int OFFXT_RGB = 0;
constexpr int OFFXT_DIR = OFFXT_RGB + FXT_RGB;
constexpr int OFFXT_FIN = OFFXT_DIR + FXT_DIR;
constexpr int OFFXT_L8 = OFFXT_FIN + FXT_HID;        // L8, L7, ..., L2 contiguous
constexpr int STREAMXT_FRAGS = OFFXT_L8 + 7 * FXT_HID;               // 3312
constexpr int STAGESXT_PER_PASS = STREAMXT_FRAGS / STAGE_FRAGS;      // 207
static_assert(STREAMXT_FRAGS % STAGE_FRAGS == 0 && FXT_RGB % STAGE_FRAGS == 0 && FXT_DIR % STAGE_FRAGS == 0 && FXT_RGB % 6 == 0 && FXT_DIR % 6 == 0,
              "transposed x3 layers must be whole stages and whole queue turns");
constexpr size_t PACKEDXT_BYTES = (size_t)CONST_BYTES + (size_t)STREAMXT_FRAGS * FRAG_BYTES;

// ---- "fragHT": the transposed stream with TWO fp16 pieces per weight, scaled by H2_WSCALE like fragH, for the backward-data pass on the h2 core
// (mlp_backward_h2.hip): fragXT's geometry and layer order with two pieces per (k-step, tile).
constexpr int FHT_RGB = (FEAT_DIM / 16) * 4 * 2;     //  32: static_rgb^T      64 -> 128
constexpr int FHT_DIR = (128 / 16) * 8 * 2;          // 128: dir_encoding^T   128 -> 256
constexpr int FHT_HID = KS_HID * 8 * 2;              // 256: xyz_encoding_final^T and xyz_encoding_{8..2}^T
constexpr int OFFHT_RGB = 0;
constexpr int OFFHT_DIR = OFFHT_RGB + FHT_RGB;
constexpr int OFFHT_FIN = OFFHT_DIR + FHT_DIR;
constexpr int OFFHT_L8 = OFFHT_FIN + FHT_HID;        // L8, L7, ..., L2 contiguous
constexpr int STREAMHT_FRAGS = OFFHT_L8 + 7 * FHT_HID;               // 2208
static_assert(STREAMHT_FRAGS % STAGE_FRAGS == 0 && FHT_RGB % STAGE_FRAGS == 0 && FHT_DIR % STAGE_FRAGS == 0, "transposed h2 layers must be whole stages");
constexpr size_t PACKEDHT_BYTES = (size_t)CONST_BYTES + (size_t)STREAMHT_FRAGS * FRAG_BYTES;

}  // namespace crnerf
