// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3) on gfx950: operand packing and scale semantics, with exact integer data.
// Build: hipcc --offload-arch=gfx950 -O2 mx_probe.hip -o mx_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
static unsigned char enc(int v) {
  if (v == 0) return 0;
  unsigned char s = v < 0 ? 0x80 : 0; int m = abs(v);
  const unsigned char t[5] = {0, 0x38, 0x40, 0x44, 0x48};
  return s | t[m];
}
static unsigned char ha[64][32], hb[64][32]; static int hsa[64], hsb[64]; static float hc[64][4];
static void *da, *db, *dsa, *dsb, *dc;
static void run() {
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (const int*)dsa, (const int*)dsb, (v4f*)dc);
  hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
}
static float C(int row, int col) { return hc[(row >> 2) * 16 + col][row & 3]; }
int main() {
  (void)hipMalloc(&da, sizeof(ha)); (void)hipMalloc(&db, sizeof(hb)); (void)hipMalloc(&dsa, sizeof(hsa)); (void)hipMalloc(&dsb, sizeof(hsb)); (void)hipMalloc(&dc, 64 * 16);
  // E1: unit scales, random small integers, hypothesis lane l: row/col l&15, k = 32 (l>>4) + byte
  float A[16][128], B[128][16];
  srand(7);
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) { A[i][kk] = rand() % 9 - 4; B[kk][i] = rand() % 9 - 4; }
  for (int l = 0; l < 64; ++l) {
    int r = l & 15, g = l >> 4;
    for (int j = 0; j < 32; ++j) { ha[l][j] = enc((int)A[r][32 * g + j]); hb[l][j] = enc((int)B[32 * g + j][r]); }
    hsa[l] = 127; hsb[l] = 127;
  }
  run();
  int bad = 0;
  for (int row = 0; row < 16; ++row) for (int col = 0; col < 16; ++col) {
    double ref = 0; for (int kk = 0; kk < 128; ++kk) ref += A[row][kk] * B[kk][col];
    if (fabs(ref - C(row, col)) > 1e-3) { if (bad < 4) printf("E1 mismatch row %d col %d: got %g want %g\n", row, col, C(row, col), ref); ++bad; }
  }
  printf("E1 (unit scales, data packing): %d mismatches of 256\n", bad);
  // E2: ones everywhere; one lane's A scale raised to 128 (x2), upper scale bytes junk
  for (int l0 : {0, 5, 16, 21, 32, 48, 63}) {
    memset(ha, 0x38, sizeof(ha)); memset(hb, 0x38, sizeof(hb));
    for (int l = 0; l < 64; ++l) { hsa[l] = 127 | 0x7f7f7f00; hsb[l] = 127 | 0x7f7f7f00; }
    hsa[l0] = 128 | 0x7f7f7f00;
    run();
    printf("E2 A-scale of lane %2d = 2: C[row][0] =", l0);
    for (int row = 0; row < 16; ++row) printf(" %g", C(row, 0));
    printf("\n");
  }
  for (int l0 : {0, 5, 16, 48}) {
    memset(ha, 0x38, sizeof(ha)); memset(hb, 0x38, sizeof(hb));
    for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
    hsb[l0] = 129;
    run();
    printf("E2 B-scale of lane %2d = 4: C[0][col] =", l0);
    for (int col = 0; col < 16; ++col) printf(" %g", C(0, col));
    printf("\n");
  }
  // E3: which bytes of a lane does its scale cover?  A = 1 only in byte j of lane 0 (row 0, group 0), B ones, lane 0 scale x2
  for (int j : {0, 1, 15, 16, 31}) {
    memset(ha, 0, sizeof(ha)); memset(hb, 0x38, sizeof(hb));
    ha[0][j] = 0x38;
    for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
    hsa[0] = 128;
    run();
    printf("E3 single A element lane 0 byte %2d, lane-0 scale 2: C[0][0] = %g, C[1][0] = %g\n", j, C(0, 0), C(1, 0));
  }
  // E4: scale byte selection: put the scale in byte 1 / 2 / 3 with byte 0 = 127
  for (int byte = 0; byte < 4; ++byte) {
    memset(ha, 0x38, sizeof(ha)); memset(hb, 0x38, sizeof(hb));
    for (int l = 0; l < 64; ++l) { hsa[l] = 0x7f7f7f7f; hsb[l] = 0x7f7f7f7f; hsa[l] = (hsa[l] & ~(0xff << (8 * byte))) | (128 << (8 * byte)); }
    run();
    printf("E4 every A scale = 2 in byte %d (others 127): C[0][0] = %g\n", byte, C(0, 0));
  }
  // E5: hypothesis H2 -- lane (r, g): bytes 0..15 are k = 16 g + j, bytes 16..31 are k = 64 + 16 g + (j - 16); the scale in byte 0 of lane
  // (r, g) applies to the MX block k in [32 g, 32 g + 32) of row / column r.  Random data AND random scales.
  {
    int SA[16][4], SB[16][4];
    for (int i = 0; i < 16; ++i) for (int g = 0; g < 4; ++g) { SA[i][g] = 127 + (rand() % 7 - 3); SB[i][g] = 127 + (rand() % 7 - 3); }
    for (int l = 0; l < 64; ++l) {
      int r = l & 15, g = l >> 4;
      for (int j = 0; j < 32; ++j) {
        int kk = j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16);
        ha[l][j] = enc((int)A[r][kk]); hb[l][j] = enc((int)B[kk][r]);
      }
      hsa[l] = SA[r][g] | 0x11223300; hsb[l] = SB[r][g] | 0x44556600;
    }
    run();
    int bad5 = 0;
    for (int row = 0; row < 16; ++row) for (int col = 0; col < 16; ++col) {
      double ref = 0;
      for (int g = 0; g < 4; ++g) { double t = 0; for (int j = 0; j < 32; ++j) t += A[row][32 * g + j] * B[32 * g + j][col]; ref += t * pow(2.0, SA[row][g] - 127 + SB[col][g] - 127); }
      if (fabs(ref - C(row, col)) > 1e-3 * (1 + fabs(ref))) { if (bad5 < 4) printf("E5 mismatch row %d col %d: got %g want %g\n", row, col, C(row, col), ref); ++bad5; }
    }
    printf("E5 (H2: split halves, block scales): %d mismatches of 256\n", bad5);
  }
  return 0;
}
