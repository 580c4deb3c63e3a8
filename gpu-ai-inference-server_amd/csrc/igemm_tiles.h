// Tile configurations of the MFMA implicit-GEMM convolution kernel, shared by the planner (choice) and
// the kernels (template instantiation).  BM x BN output tile per workgroup, WM x WN waves (64 lanes each),
// every wave owns a (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA blocks (v_mfma_f32_32x32x2_f32).
#pragma once

namespace ie {

struct IgemmTile { int bm, bn, wm, wn; };

constexpr int kNumIgemmTiles = 7;
constexpr IgemmTile kIgemmTiles[kNumIgemmTiles] = {
    {128, 128, 2, 2},   // 0: wave 64x64  (1x1 bottlenecks, big M)
    {128, 64, 2, 2},    // 1: wave 64x32
    {128, 32, 4, 1},    // 2: wave 32x32  (3x3 growth convs, Cout = 32)
    {64, 64, 2, 2},     // 3: wave 32x32  (mid M)
    {64, 32, 2, 1},     // 4: 128 threads (small M)
    {32, 32, 1, 1},     // 5: 64 threads  (tiny M)
    {256, 32, 4, 1},    // 6: wave 64x32  (3x3 growth convs, very large M)
};
constexpr int kIgemmBK = 32;       // K-tile depth (floats)
constexpr int kIgemmLdsPad = 4;    // row pitch = BK + 4 floats: conflict-free ds_read_b128 (pitch/4 odd)

}  // namespace ie
