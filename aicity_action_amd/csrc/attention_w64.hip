// Fused pooled attention forward, head_dim 96, 64 queries per wave (4 waves = 256 queries per workgroup, ONE wave per SIMD).
//
// The timing ablations of the 32-query kernels show that the K/V fragment reads from LDS, the LDS-DMA
// and the MFMAs cost their own time one after the other; with two 32-query blocks per wave every K / V fragment read from
// LDS feeds TWO MFMAs and a K/V tile is streamed once per 256 queries, so both per-MFMA costs halve.  The price is the
// register file: S^T (64 keys x 64 queries, 64 registers), P (32), Q^T (48) and the fragments live in the arch VGPRs, the six
// O^T accumulator tiles (96 registers) in the ACC registers (MFMA issued as inline asm with "a" operands), 1 wave per SIMD.
#include "common.h"

#define W_KT 64           // keys per tile
#define W_ROWB 192        // bytes per K/V row in LDS (96 bf16)
#define W_TILE (W_KT * W_ROWB)           // one K or V tile image: 12 KiB
#define W_STAGES 3
#define W_QB 256          // queries per workgroup

typedef __attribute__((ext_vector_type(2))) float wf32x2;

// ---- ACC-register tiles owned by inline asm: O^T tile t = 3*(query block) + (32-d block) lives in a[16t : 16t+15] ----
#define W_ACC0 "a[0:15]"
#define W_CLOB0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define W_ACC1 "a[16:31]"
#define W_CLOB1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define W_ACC2 "a[32:47]"
#define W_CLOB2 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define W_ACC3 "a[48:63]"
#define W_CLOB3 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define W_ACC4 "a[64:79]"
#define W_CLOB4 "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79"
#define W_ACC5 "a[80:95]"
#define W_CLOB5 "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define W_ZERO_ALL "v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0"
#define W_CLOB_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define W_SCALE0 "s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a1\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_read_b32 %0, a2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_read_b32 %0, a3\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a3, %0\n\tv_accvgpr_read_b32 %0, a4\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a4, %0\n\tv_accvgpr_read_b32 %0, a5\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_read_b32 %0, a6\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_read_b32 %0, a7\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a7, %0\n\tv_accvgpr_read_b32 %0, a8\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_read_b32 %0, a9\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a9, %0\n\tv_accvgpr_read_b32 %0, a10\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_read_b32 %0, a11\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a11, %0\n\tv_accvgpr_read_b32 %0, a12\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a12, %0\n\tv_accvgpr_read_b32 %0, a13\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a13, %0\n\tv_accvgpr_read_b32 %0, a14\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_read_b32 %0, a15\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a15, %0\n\tv_accvgpr_read_b32 %0, a16\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a16, %0\n\tv_accvgpr_read_b32 %0, a17\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a17, %0\n\tv_accvgpr_read_b32 %0, a18\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_read_b32 %0, a19\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a19, %0\n\tv_accvgpr_read_b32 %0, a20\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a20, %0\n\tv_accvgpr_read_b32 %0, a21\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a21, %0\n\tv_accvgpr_read_b32 %0, a22\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_read_b32 %0, a23\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a23, %0\n\tv_accvgpr_read_b32 %0, a24\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a24, %0\n\tv_accvgpr_read_b32 %0, a25\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a25, %0\n\tv_accvgpr_read_b32 %0, a26\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a26, %0\n\tv_accvgpr_read_b32 %0, a27\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a27, %0\n\tv_accvgpr_read_b32 %0, a28\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a28, %0\n\tv_accvgpr_read_b32 %0, a29\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a29, %0\n\tv_accvgpr_read_b32 %0, a30\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a30, %0\n\tv_accvgpr_read_b32 %0, a31\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a31, %0\n\tv_accvgpr_read_b32 %0, a32\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a32, %0\n\tv_accvgpr_read_b32 %0, a33\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_read_b32 %0, a34\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_read_b32 %0, a35\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a35, %0\n\tv_accvgpr_read_b32 %0, a36\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a36, %0\n\tv_accvgpr_read_b32 %0, a37\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a37, %0\n\tv_accvgpr_read_b32 %0, a38\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_read_b32 %0, a39\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a39, %0\n\tv_accvgpr_read_b32 %0, a40\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a40, %0\n\tv_accvgpr_read_b32 %0, a41\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a41, %0\n\tv_accvgpr_read_b32 %0, a42\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a42, %0\n\tv_accvgpr_read_b32 %0, a43\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a43, %0\n\tv_accvgpr_read_b32 %0, a44\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a44, %0\n\tv_accvgpr_read_b32 %0, a45\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a45, %0\n\tv_accvgpr_read_b32 %0, a46\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a46, %0\n\tv_accvgpr_read_b32 %0, a47\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a47, %0\n\ts_nop 7"
#define W_CLOBQ0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define W_SCALE1 "s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a48\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a48, %0\n\tv_accvgpr_read_b32 %0, a49\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a49, %0\n\tv_accvgpr_read_b32 %0, a50\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a50, %0\n\tv_accvgpr_read_b32 %0, a51\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a51, %0\n\tv_accvgpr_read_b32 %0, a52\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a52, %0\n\tv_accvgpr_read_b32 %0, a53\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a53, %0\n\tv_accvgpr_read_b32 %0, a54\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a54, %0\n\tv_accvgpr_read_b32 %0, a55\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a55, %0\n\tv_accvgpr_read_b32 %0, a56\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a56, %0\n\tv_accvgpr_read_b32 %0, a57\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a57, %0\n\tv_accvgpr_read_b32 %0, a58\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a58, %0\n\tv_accvgpr_read_b32 %0, a59\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a59, %0\n\tv_accvgpr_read_b32 %0, a60\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a60, %0\n\tv_accvgpr_read_b32 %0, a61\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a61, %0\n\tv_accvgpr_read_b32 %0, a62\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a62, %0\n\tv_accvgpr_read_b32 %0, a63\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a63, %0\n\tv_accvgpr_read_b32 %0, a64\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a64, %0\n\tv_accvgpr_read_b32 %0, a65\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a65, %0\n\tv_accvgpr_read_b32 %0, a66\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_read_b32 %0, a67\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a67, %0\n\tv_accvgpr_read_b32 %0, a68\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a68, %0\n\tv_accvgpr_read_b32 %0, a69\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a69, %0\n\tv_accvgpr_read_b32 %0, a70\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a70, %0\n\tv_accvgpr_read_b32 %0, a71\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a71, %0\n\tv_accvgpr_read_b32 %0, a72\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a72, %0\n\tv_accvgpr_read_b32 %0, a73\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a73, %0\n\tv_accvgpr_read_b32 %0, a74\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a74, %0\n\tv_accvgpr_read_b32 %0, a75\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a75, %0\n\tv_accvgpr_read_b32 %0, a76\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a76, %0\n\tv_accvgpr_read_b32 %0, a77\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a77, %0\n\tv_accvgpr_read_b32 %0, a78\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a78, %0\n\tv_accvgpr_read_b32 %0, a79\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a79, %0\n\tv_accvgpr_read_b32 %0, a80\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a80, %0\n\tv_accvgpr_read_b32 %0, a81\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a81, %0\n\tv_accvgpr_read_b32 %0, a82\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a82, %0\n\tv_accvgpr_read_b32 %0, a83\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a83, %0\n\tv_accvgpr_read_b32 %0, a84\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a84, %0\n\tv_accvgpr_read_b32 %0, a85\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a85, %0\n\tv_accvgpr_read_b32 %0, a86\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a86, %0\n\tv_accvgpr_read_b32 %0, a87\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a87, %0\n\tv_accvgpr_read_b32 %0, a88\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a88, %0\n\tv_accvgpr_read_b32 %0, a89\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a89, %0\n\tv_accvgpr_read_b32 %0, a90\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a90, %0\n\tv_accvgpr_read_b32 %0, a91\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a91, %0\n\tv_accvgpr_read_b32 %0, a92\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a92, %0\n\tv_accvgpr_read_b32 %0, a93\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a93, %0\n\tv_accvgpr_read_b32 %0, a94\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a94, %0\n\tv_accvgpr_read_b32 %0, a95\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a95, %0\n\ts_nop 7"
#define W_CLOBQ1 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"

template <int OFF>
__device__ __forceinline__ bf16x4 w_tr16(uint32_t addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 w_rd128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
#ifdef MVIT_HALF_IS_FP16
#define W_MFMA_OP "v_mfma_f32_32x32x16_f16 "
#else
#define W_MFMA_OP "v_mfma_f32_32x32x16_bf16 "
#endif
#define W_PVMFMA(T, VF, PF) asm volatile(W_MFMA_OP W_ACC##T ", %0, %1, " W_ACC##T ::"v"(VF), "v"(PF) : W_CLOB##T)

template <bool ADD_Q>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kt,
                                                              const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                              float* __restrict__ LSE, int heads, int Lq, int Lk,
                                                              float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // W_STAGES x (K image 12 KiB | V image 12 KiB)

    int qtile, bh;
    xcd_group_map(qtile, bh);
    const int b = bh / heads, g = bh - b * heads;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qtile * W_QB + wave * 64;

    const bf16_t* Qb = Q + (int64_t)bh * Lq * 96;
    const bf16_t* Kb = Kt + (int64_t)bh * Lk * 96;
    const bf16_t* Vb = V + (int64_t)bh * Lk * 96;

    // Q^T fragments of the two 32-query blocks: lane (r,h) holds Q[q0 + 32j + r][16ks + 8h .. +7]
    int qi[2];
    bool q_ok[2];
    bf16x8 qf[2][6];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qi[j] = q0 + 32 * j + r;
        q_ok[j] = qi[j] < Lq;
        qi[j] = q_ok[j] ? qi[j] : Lq - 1;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) qf[j][ks] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qi[j] * 96 + 16 * ks + 8 * h);
    }

    // DMA: waves 0,1 move the K image, waves 2,3 the V image of a tile, 6 one-KiB pieces each (see attention.hip)
    const bool is_v = wave >= 2;
    const char* src_bh = reinterpret_cast<const char*>(is_v ? Vb : Kb);
    auto piece_off = [&](int i, int ln, int last_row) -> uint32_t {
        const int p = 64 * (6 * (wave & 1) + i) + ln;
        int row = p / 12, c = p - row * 12;
        if (!is_v) {
            c -= (row >> 2) & 3;
            c = c < 0 ? c + 12 : c;
        }
        row = row < last_row ? row : last_row;
        return (uint32_t)(row * 12 + c) * 16u;
    };
    uint32_t g_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_off[i] = piece_off(i, lane, W_KT);
    const uint32_t smem_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem);
    const uint32_t ring_a = smem_a + (is_v ? W_TILE : 0) + 1024 * (6 * (wave & 1));
    auto dma1 = [&](const char* base, uint32_t off, uint32_t lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
    };
    auto dma = [&](int tile, int stage) {
        const int k0 = tile * W_KT;
        const char* t_base = src_bh + (int64_t)k0 * W_ROWB;     // wave-uniform
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_a + stage * (2 * W_TILE));
        if (k0 + W_KT <= Lk) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                dma1(t_base, g_off[i], dst + 1024 * i);
                dma1(t_base + 16 * W_ROWB, g_off[i], dst + 1024 * (i + 3));     // pieces i and i+3 are 16 rows apart
            }
        } else {        // tail tile: rows past Lk re-read the last valid row (finite data; their scores are masked to -inf)
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int i = 0; i < 6; ++i) dma1(t_base, piece_off(i, ln, Lk - 1 - k0), dst + 1024 * i);
        }
    };

    uint32_t ka0, ka4, ka5;
    {
        const int p0 = h + ((r >> 2) & 3);
        const int p4 = p0 + 8 >= 12 ? p0 + 8 - 12 : p0 + 8, p5 = p0 + 10 >= 12 ? p0 + 10 - 12 : p0 + 10;
        ka0 = smem_a + r * W_ROWB + p0 * 16;
        ka4 = smem_a + r * W_ROWB + p4 * 16;
        ka5 = smem_a + r * W_ROWB + p5 * 16;
    }
    const int i16 = lane & 15, gi = lane >> 4;
    const uint32_t va0 = smem_a + W_TILE + (4 * h + (i16 >> 2)) * W_ROWB + (16 * (gi & 1) + 4 * (i16 & 3)) * 2;

    asm volatile(W_ZERO_ALL ::: W_CLOB_ALL);      // O^T = 0 (six ACC tiles)
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    const int nkt = (Lk + W_KT - 1) / W_KT;
    dma(0, 0);
    if (nkt > 1) dma(1, 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(qf[j][ks]));
    int stage = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) dma(kt + 2, stage == 0 ? 2 : stage - 1);
        const uint32_t sb = stage * (2 * W_TILE);
        stage = stage == W_STAGES - 1 ? 0 : stage + 1;

        // ---- S^T = K . Q^T for both query blocks: every K fragment feeds two MFMAs ----------------------
        f32x16 s[2][2];       // [query block][32-key block]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[j][kb][i] = 0.f;
        {
            bf16x8 kf[6][2];
            const uint32_t a0 = ka0 + sb, a4 = ka4 + sb, a5 = ka5 + sb;
#define WK(KS, A, OFF) kf[KS][0] = w_rd128<OFF>(A); kf[KS][1] = w_rd128<OFF + 32 * W_ROWB>(A);
            WK(0, a0, 0) WK(1, a0, 32) WK(2, a0, 64) WK(3, a0, 96) WK(4, a4, 0) WK(5, a5, 0)
#undef WK
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[2][0]), "+v"(kf[2][1]));
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) s[j][kb] = mfma16(kf[ks][kb], qf[j][ks], s[j][kb]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[3][0]), "+v"(kf[3][1]), "+v"(kf[4][0]), "+v"(kf[4][1]), "+v"(kf[5][0]), "+v"(kf[5][1]));
#pragma unroll
            for (int ks = 3; ks < 6; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) s[j][kb] = mfma16(kf[ks][kb], qf[j][ks], s[j][kb]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // V^T fragments of the first 32 keys: requested now, they land under the softmax arithmetic
        bf16x4 vl[6], vh[6];
        const uint32_t va = va0 + sb;
#define WV(S16, DB) vl[3 * (S16 & 1) + DB] = w_tr16<S16 * 16 * W_ROWB + DB * 64>(va); vh[3 * (S16 & 1) + DB] = w_tr16<S16 * 16 * W_ROWB + DB * 64 + 8 * W_ROWB>(va);
        WV(0, 0) WV(0, 1) WV(0, 2) WV(1, 0) WV(1, 1) WV(1, 2)

        // ---- online softmax per query block ---------------------------------------------------------------
        const int kbase = kt * W_KT;
        bf16x8 pf[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (kbase + W_KT > Lk) {           // tail tile only (wave-uniform): mask keys >= Lk
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kbase + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                        s[j][kb][i] = key < Lk ? s[j][kb][i] : -INFINITY;
                    }
            }
            float mx = fmaxf(s[j][0][0], s[j][1][0]);
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s[j][0][i]), s[j][1][i]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float m_new = fmaxf(m_run[j], mx);
            if (__any(m_new > m_run[j])) {        // rescale only when some query's running max moved
                const float alpha = __builtin_amdgcn_exp2f((m_run[j] - m_new) * scale_log2e);
                l_run[j] *= alpha;
                float tmp_;
                if (j == 0) asm volatile(W_SCALE0 : "=&v"(tmp_) : "v"(alpha) : W_CLOBQ0);
                else asm volatile(W_SCALE1 : "=&v"(tmp_) : "v"(alpha) : W_CLOBQ1);
                m_run[j] = m_new;
            }
            const wf32x2 c2 = {scale_log2e, scale_log2e};
            const float mcs = -m_run[j] * scale_log2e;
            const wf32x2 mc2 = {mcs, mcs};
            wf32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    uint32_t pk[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const wf32x2 sv = {s[j][kb][8 * sh + 2 * jj], s[j][kb][8 * sh + 2 * jj + 1]};
                        const wf32x2 tt = __builtin_elementwise_fma(sv, c2, mc2);
                        const wf32x2 pp = {__builtin_amdgcn_exp2f(tt[0]), __builtin_amdgcn_exp2f(tt[1])};
                        ps2 += pp;
                        pk[jj] = pack_bf16x2(pp[0], pp[1]);
                    }
                    uint4 u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    pf[j][2 * kb + sh] = *reinterpret_cast<bf16x8*>(&u);
                }
            l_run[j] += ps2[0] + ps2[1];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T += V^T . P^T: every V^T fragment feeds two MFMAs (accumulators in ACC registers) ---------
#define WPV1(S16, DB, T0, T1) { \
            const bf16x4 lo_ = vl[3 * (S16 & 1) + DB], hi_ = vh[3 * (S16 & 1) + DB]; \
            bf16x8 vf_; \
            vf_[0] = lo_[0]; vf_[1] = lo_[1]; vf_[2] = lo_[2]; vf_[3] = lo_[3]; vf_[4] = hi_[0]; vf_[5] = hi_[1]; vf_[6] = hi_[2]; vf_[7] = hi_[3]; \
            W_PVMFMA(T0, vf_, pf[0][S16]); \
            W_PVMFMA(T1, vf_, pf[1][S16]); }
#define WPV(S16) WPV1(S16, 0, 0, 3) WPV1(S16, 1, 1, 4) WPV1(S16, 2, 2, 5)
#define WWAIT(N, A, B, C, D, E, F) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E), "+v"(F) : "n"(N))
        WWAIT(6, vl[0], vl[1], vl[2], vh[0], vh[1], vh[2]);
        WPV(0)
        WV(2, 0) WV(2, 1) WV(2, 2)
        WWAIT(6, vl[3], vl[4], vl[5], vh[3], vh[4], vh[5]);
        WPV(1)
        WV(3, 0) WV(3, 1) WV(3, 2)
        WWAIT(6, vl[0], vl[1], vl[2], vh[0], vh[1], vh[2]);
        WPV(2)
        WWAIT(0, vl[3], vl[4], vl[5], vh[3], vh[4], vh[5]);
        WPV(3)
#undef WPV
#undef WPV1
#undef WV
#undef WWAIT
    }

    // ---- epilogue: normalise, + q residual, store [b][q][g*96 + d] -------------------------------
    float ot[6][16];
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last MFMAs have left the matrix pipe before the ACC reads
    asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\tv_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\tv_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15" : "=v"(ot[0][0]), "=v"(ot[0][1]), "=v"(ot[0][2]), "=v"(ot[0][3]), "=v"(ot[0][4]), "=v"(ot[0][5]), "=v"(ot[0][6]), "=v"(ot[0][7]), "=v"(ot[0][8]), "=v"(ot[0][9]), "=v"(ot[0][10]), "=v"(ot[0][11]), "=v"(ot[0][12]), "=v"(ot[0][13]), "=v"(ot[0][14]), "=v"(ot[0][15]));
    asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\tv_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\tv_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\tv_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31" : "=v"(ot[1][0]), "=v"(ot[1][1]), "=v"(ot[1][2]), "=v"(ot[1][3]), "=v"(ot[1][4]), "=v"(ot[1][5]), "=v"(ot[1][6]), "=v"(ot[1][7]), "=v"(ot[1][8]), "=v"(ot[1][9]), "=v"(ot[1][10]), "=v"(ot[1][11]), "=v"(ot[1][12]), "=v"(ot[1][13]), "=v"(ot[1][14]), "=v"(ot[1][15]));
    asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\tv_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\tv_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\tv_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\tv_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\tv_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\tv_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47" : "=v"(ot[2][0]), "=v"(ot[2][1]), "=v"(ot[2][2]), "=v"(ot[2][3]), "=v"(ot[2][4]), "=v"(ot[2][5]), "=v"(ot[2][6]), "=v"(ot[2][7]), "=v"(ot[2][8]), "=v"(ot[2][9]), "=v"(ot[2][10]), "=v"(ot[2][11]), "=v"(ot[2][12]), "=v"(ot[2][13]), "=v"(ot[2][14]), "=v"(ot[2][15]));
    asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\tv_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\tv_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\tv_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\tv_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\tv_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63" : "=v"(ot[3][0]), "=v"(ot[3][1]), "=v"(ot[3][2]), "=v"(ot[3][3]), "=v"(ot[3][4]), "=v"(ot[3][5]), "=v"(ot[3][6]), "=v"(ot[3][7]), "=v"(ot[3][8]), "=v"(ot[3][9]), "=v"(ot[3][10]), "=v"(ot[3][11]), "=v"(ot[3][12]), "=v"(ot[3][13]), "=v"(ot[3][14]), "=v"(ot[3][15]));
    asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\tv_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\tv_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\tv_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\tv_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\tv_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79" : "=v"(ot[4][0]), "=v"(ot[4][1]), "=v"(ot[4][2]), "=v"(ot[4][3]), "=v"(ot[4][4]), "=v"(ot[4][5]), "=v"(ot[4][6]), "=v"(ot[4][7]), "=v"(ot[4][8]), "=v"(ot[4][9]), "=v"(ot[4][10]), "=v"(ot[4][11]), "=v"(ot[4][12]), "=v"(ot[4][13]), "=v"(ot[4][14]), "=v"(ot[4][15]));
    asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83\n\tv_accvgpr_read_b32 %4, a84\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a86\n\tv_accvgpr_read_b32 %7, a87\n\tv_accvgpr_read_b32 %8, a88\n\tv_accvgpr_read_b32 %9, a89\n\tv_accvgpr_read_b32 %10, a90\n\tv_accvgpr_read_b32 %11, a91\n\tv_accvgpr_read_b32 %12, a92\n\tv_accvgpr_read_b32 %13, a93\n\tv_accvgpr_read_b32 %14, a94\n\tv_accvgpr_read_b32 %15, a95" : "=v"(ot[5][0]), "=v"(ot[5][1]), "=v"(ot[5][2]), "=v"(ot[5][3]), "=v"(ot[5][4]), "=v"(ot[5][5]), "=v"(ot[5][6]), "=v"(ot[5][7]), "=v"(ot[5][8]), "=v"(ot[5][9]), "=v"(ot[5][10]), "=v"(ot[5][11]), "=v"(ot[5][12]), "=v"(ot[5][13]), "=v"(ot[5][14]), "=v"(ot[5][15]));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
        const float inv = 1.0f / l_tot;
        if (LSE && q_ok[j] && h == 0) LSE[(int64_t)bh * Lq + qi[j]] = m_run[j] * scale_log2e + __builtin_amdgcn_logf(l_tot);  // log2 domain
        if (q_ok[j]) {
            const int C = heads * 96;
            bf16_t* orow = O + ((int64_t)b * Lq + qi[j]) * C + g * 96;
            const bf16_t* qrow = Qb + (int64_t)qi[j] * 96;
#pragma unroll
            for (int db = 0; db < 3; ++db)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int d = 32 * db + 8 * i4 + 4 * h;
                    const float* oo = ot[3 * j + db];
                    float4 v = make_float4(oo[4 * i4 + 0] * inv, oo[4 * i4 + 1] * inv, oo[4 * i4 + 2] * inv, oo[4 * i4 + 3] * inv);
                    if (ADD_Q) {
                        const float4 qq = load4(qrow + d);
                        v.x += qq.x; v.y += qq.y; v.z += qq.z; v.w += qq.w;
                    }
                    store4(orow + d, v);
                }
        }
    }
}

// launcher used by mvit_attention_fwd (attention.hip) when this form is selected
int attn_fwd_w64_launch(const void* q, const void* k, const void* v, void* out, float* lse, int B, int heads, int Lq, int Lk,
                        float scale_log2e, int add_q, hipStream_t st) {
    dim3 grid((Lq + W_QB - 1) / W_QB, B * heads);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_w64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W_STAGES * 2 * W_TILE) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_w64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_STAGES * 2 * W_TILE) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    if (add_q) hipLaunchKernelGGL((attn_fwd_w64_kernel<true>), grid, dim3(256), W_STAGES * 2 * W_TILE, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, lse, heads, Lq, Lk, scale_log2e);
    else hipLaunchKernelGGL((attn_fwd_w64_kernel<false>), grid, dim3(256), W_STAGES * 2 * W_TILE, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, lse, heads, Lq, Lk, scale_log2e);
    return MVIT_OK;
}
