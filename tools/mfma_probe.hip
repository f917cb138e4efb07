// Probe: operand / result layout of v_mfma_i32_4x4x4_16b_i8 on gfx950 (the colour-matrix experiment of mx_k_video.hip, matrix mode 4).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* a, const int* b, i32x4* d) {
    const int l = threadIdx.x;
    i32x4 c = {1000, 2000, 3000, 4000};
    d[l] = __builtin_amdgcn_mfma_i32_4x4x4i8(a[l], b[l], c, 0, 0, 0);
}
int main() {
    int ha[64], hb[64]; i32x4 hd[64];
    auto pk = [](int x0, int x1, int x2, int x3) { return (int)((uint32_t)(uint8_t)x0 | ((uint32_t)(uint8_t)x1 << 8) | ((uint32_t)(uint8_t)x2 << 16) | ((uint32_t)(uint8_t)x3 << 24)); };
    int A[16][4][4], B[16][4][4];   // [block][row i][k], [block][k][col j]
    for (int blk = 0; blk < 16; ++blk) for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) { A[blk][i][k] = (blk * 7 + i * 5 + k * 3) % 23 - 11; B[blk][k][i] = (blk * 3 + i * 11 + k * 2) % 19 - 9; }
    for (int l = 0; l < 64; ++l) { const int blk = l / 4, i = l % 4; ha[l] = pk(A[blk][i][0], A[blk][i][1], A[blk][i][2], A[blk][i][3]); hb[l] = pk(B[blk][0][i], B[blk][1][i], B[blk][2][i], B[blk][3][i]); }
    int *da, *db; i32x4* dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad_ij = 0, bad_ji = 0;
    for (int l = 0; l < 64; ++l) { const int blk = l / 4, j = l % 4; const int got[4] = {hd[l].x, hd[l].y, hd[l].z, hd[l].w};
        for (int i = 0; i < 4; ++i) { int s = 1000 * (i + 1), t = 1000 * (i + 1); for (int k = 0; k < 4; ++k) { s += A[blk][i][k] * B[blk][k][j]; t += A[blk][j][k] * B[blk][k][i]; } bad_ij += got[i] != s; bad_ji += got[i] != t; } }
    printf("layout 'lane (block, j) register i = D[i][j] = sum_k A[i][k] B[k][j], A row from lane (block, i), B column from the lane itself': %s (%d mismatches; transposed reading: %d)\n", bad_ij ? "NO" : "YES", bad_ij, bad_ji);
    return bad_ij ? 1 : 0;
}
