// Tile configurations of the MFMA implicit-GEMM convolution kernel, shared by the planner (choice) and
// the kernels (template instantiation).  BM x BN output tile per workgroup, WM x WN waves (64 lanes each),
// every wave owns a (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA blocks (v_mfma_f32_32x32x2_f32).
#pragma once

namespace ie {

struct IgemmTile { int bm, bn, wm, wn, kg, deep; };   // kg: K-groups inside the workgroup; deep: 2 K-tiles of register prefetch

constexpr int kNumIgemmTiles = 16;
constexpr int kNumIgemmBaseTiles = 7;   // tiles 0..6 have kg == 1 (the planner's heuristic only picks among these)
constexpr IgemmTile kIgemmTiles[kNumIgemmTiles] = {
    {128, 128, 2, 2, 1, 0},   // 0: wave 64x64  (1x1 bottlenecks, big M)
    {128, 64, 2, 2, 1, 0},    // 1: wave 64x32
    {128, 32, 4, 1, 1, 0},    // 2: wave 32x32  (3x3 growth convs, Cout = 32)
    {64, 64, 2, 2, 1, 0},     // 3: wave 32x32  (mid M)
    {64, 32, 2, 1, 1, 0},     // 4: 128 threads (small M)
    {32, 32, 1, 1, 1, 0},     // 5: 64 threads  (tiny M)
    {256, 32, 4, 1, 1, 0},    // 6: wave 64x32  (3x3 growth convs, very large M)
    // Small-M layers (dense blocks 3-4): the output grid cannot fill 1024 SIMDs, so 2 or 4 wave groups of ONE workgroup
    // each take a slice of K with their own LDS staging buffers and the partial tiles are summed through LDS at the end
    // (no global slabs, no second kernel).
    {64, 64, 2, 2, 2, 0},     // 7: 512 threads
    {64, 64, 2, 2, 4, 0},     // 8: 1024 threads
    {32, 32, 1, 1, 4, 0},     // 9: 256 threads
    {128, 64, 2, 2, 2, 0}, // 10: 512 threads
    // Latency-bound variants for the same small layers: with one or two waves per SIMD nothing else hides the L2 round trip
    // of the next K-tile, so TWO K-tiles are kept in flight in registers (one workgroup per tile, no persistence).
    {64, 64, 2, 2, 1, 1},  // 11
    {64, 64, 2, 2, 2, 1},  // 12
    {64, 64, 2, 2, 4, 1},  // 13
    {32, 32, 1, 1, 4, 1},  // 14
    {64, 32, 2, 1, 1, 1},  // 15
};
constexpr int kIgemmBK = 32;       // K-tile depth (floats)
constexpr int kIgemmLdsPad = 4;    // row pitch = BK + 4 floats: conflict-free ds_read_b128 (pitch/4 odd)

}  // namespace ie
