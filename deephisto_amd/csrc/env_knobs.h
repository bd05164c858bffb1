// Every environment variable libdeephisto_hip.so reads, in ONE table (round 5).
//
// These are A/B and tuning switches of the engines: none of them changes WHAT is computed beyond float summation order, and every
// one has a default that is the shipped configuration.  Timing-only ablations that destroy results (DH_T2_ABL, DH_ABL, SP_ABL) and
// the gradient dump (DH_TRAIN_DUMP) are compile-time macros of diagnostic builds (tools/build_variant.sh) and do not exist in the
// shipped library (tests/test_abi.py::test_no_ablation_switch_in_shipped_library).
//
// dh::env_int(name) returns the variable's value, or the default when it is unset.  A value that is not a decimal integer inside
// [lo, hi] is NOT silently atoi'd: the default is used, one line goes to stderr, and every create entry point (dh_resnet18_create,
// dh_train2_create) fails with DH_EINVAL and dh_last_error() naming the variable (dh::env_check validates the whole table eagerly).
// INTEGRATION.md lists the same table; tests/test_abi.py compares the two through dh_debug_env_knobs.
//
//   X(name, default, lo, hi, read, effect)      read: "load" = once per process (first use), "create" = when an engine is created
#pragma once

#define DH_ENV_KNOBS(X)                                                                                                                  \
  X(DH_CONV_S2_WIDE, 1, 0, 1, "load", "bf16 inference: 1 = wide stride-2 + downsample kernel (128 couts x 256 pixels), 0 = 128-pixel kernel")      \
  X(DH_CONV_FIT, 1, 0, 1, "load", "3x3 stride-1 convs on 7x7 / 14x14 maps: 1 = fit tiles (ten 7x7 images / five 7x14 half-images per 512 slots), 0 = power-of-two tiles")   \
  X(DH_T1_SIDE, 1, 0, 1, "create", "float32 training engine: weight gradients on a low-priority side stream")                                     \
  X(DH_WGRAD3_WGS, 0, 0, 256, "load", "float32 3x3 weight gradient: workgroups per launch (0 = 224 with the side stream, else 256)")              \
  X(DH_T2_SIDE, 1, 0, 1, "create", "bf16 training engine: weight gradients, Adam and re-pack on a low-priority side stream")                       \
  X(DH_T2_JOIN, 1, 0, 1, "create", "bf16 engine: downsample branch's BN applied inside the join BN's pass (bit-identical either way)")            \
  X(DH_T2_FOLD, 1, 0, 1, "create", "bf16 engine: BN finalize folded into channel-sliced consumers on small maps (bit-identical either way)")      \
  X(DH_T2_FOLD_ROWS, 16384, 0, 1 << 24, "load", "bf16 engine: largest map (rows = B*H*W) that takes the folded BN path")                          \
  X(DH_G2_NSTAGE, 0, 0, 3, "load", "bf16 1x1 GEMM: LDS ring depth (0 = automatic: 2, or 3 from K >= DH_G2_NS3_K)")                                 \
  X(DH_G2_NS3_K, 512, 64, 1 << 20, "load", "bf16 1x1 GEMM: K from which the ring runs three stages deep")                                          \
  X(DH_G2_WIDE_M, 0, 0, 1 << 30, "load", "bf16 1x1 GEMM: 128 x 128 tiles for GEMMs of up to this many pixels (0 = off)")                          \
  X(DH_WGRAD_RING, 1, 0, 1, "load", "bf16 1x1 weight gradient: LDS-DMA ring kernel (0 = round-3 register-staged kernel)")                          \
  X(DH_WGRAD_RING_WGS, 0, 0, 1024, "load", "bf16 ring weight gradient: workgroups per launch (0 = 128 with the side stream, else 256)")           \
  X(DH_WGRAD2_WGS, 0, 0, 256, "load", "bf16 3x3 / 64-channel weight gradient: workgroups per launch (0 = 224 with the side stream, else 256)")
