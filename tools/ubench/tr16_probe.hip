// tools/exp/tr16_probe.hip — what does ds_read_b64_tr_b16 return?  (round 5, before tools/exp/attn_v7.h is trusted)
//   hipcc --offload-arch=gfx950 -O2 tools/exp/tr16_probe.hip -o gpurun_out/tr16_probe && gpurun_out/tr16_probe
// LDS holds 16-bit values equal to their own element index.  Test 1: lane l supplies the byte address 8 l (the canonical contiguous block) and
// the program prints which element index arrives in (lane, j) — the guide states (l & 15) + 16 j + 64 (l >> 4).  Test 2: the hypothesis
// attn_fwd_kernel7 is built on, with SCATTERED addresses: inside a group of 16 lanes, lane i's address supplies row (i >> 2), columns
// 4 (i & 3) .. + 3 of a 4 x 16 block, and lane c receives column c — checked with every lane's 8 bytes at an unrelated place.  Test 3: the
// kernel's own address pattern (vtr[r][dt] + the swizzle of the K / V tile image) on a tile whose element (key, d) holds key * 64 + d: lane
// (l31, hi) must receive V[key][d = 32 dt + l31] for keys 8 r + 4 hi + 0..3 of the 16-key group.  Prints PASS / FAIL per test; exit code =
// number of failed tests.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 raw4;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int* addr, int n_elems, const unsigned short* image, unsigned short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  for (int i = threadIdx.x; i < n_elems; i += 64) lds[i] = image[i];
  __syncthreads();
  const char* base = (const char*)lds;
  const raw4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) raw4*)(base + addr[threadIdx.x]));
  const u16x4 u = __builtin_bit_cast(u16x4, v);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = u[j];
}

static std::vector<unsigned short> run(const std::vector<int>& addr, const std::vector<unsigned short>& image) {
  int* d_addr; unsigned short *d_img, *d_out;
  (void)hipMalloc(&d_addr, 64 * 4); (void)hipMalloc(&d_img, image.size() * 2); (void)hipMalloc(&d_out, 256 * 2);
  (void)hipMemcpy(d_addr, addr.data(), 64 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_img, image.data(), image.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), image.size() * 2, 0, d_addr, (int)image.size(), d_img, d_out);
  std::vector<unsigned short> out(256);
  (void)hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
  (void)hipFree(d_addr); (void)hipFree(d_img); (void)hipFree(d_out);
  return out;
}

int main() {
  int failed = 0;
  {   // ---- test 1: contiguous addresses
    std::vector<unsigned short> img(4096);
    for (int i = 0; i < 4096; ++i) img[i] = (unsigned short)i;
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) addr[l] = 8 * l;
    auto out = run(addr, img);
    bool ok = true;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) ok &= out[l * 4 + j] == (l & 15) + 16 * j + 64 * (l >> 4);
    printf("test 1 (addr = 8 l; expect elem (l & 15) + 16 j + 64 (l >> 4)): %s\n", ok ? "PASS" : "FAIL");
    if (!ok) {
      ++failed;
      for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    }
  }
  {   // ---- test 2: scattered addresses, the per-group 4 x 16 block hypothesis
    std::vector<unsigned short> img(8192);
    for (int i = 0; i < 8192; ++i) img[i] = (unsigned short)i;
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) addr[l] = 8 * ((l * 37 + 11) % 1024);      // any 8-byte slot, all different
    auto out = run(addr, img);
    bool ok = true;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, c = l & 15;
        const int src = g * 16 + 4 * j + (c >> 2);       // the lane of the group that supplies row j, columns 4 (c >> 2) .. + 3
        const int want = addr[src] / 2 + (c & 3);
        ok &= out[l * 4 + j] == want;
      }
    printf("test 2 (scattered: lane c of a group gets element (c & 3) of the 8 bytes lane 4 j + (c >> 2) points at): %s\n", ok ? "PASS" : "FAIL");
    if (!ok) {
      ++failed;
      for (int l = 0; l < 64; ++l) printf("  lane %2d (addr elem %4d): %4d %4d %4d %4d\n", l, addr[l] / 2, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    }
  }
  {   // ---- test 3: attn_fwd_kernel7's address pattern on the swizzled [64 keys][64 d] tile
    std::vector<unsigned short> img(4096);
    for (int key = 0; key < 64; ++key)
      for (int d = 0; d < 64; ++d) {
        const int chunk = d >> 3;
        img[(key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4) + (d & 7) * 2) / 2] = (unsigned short)(key * 64 + d);
      }
    bool ok = true;
    for (int r = 0; r < 2; ++r)
      for (int dt = 0; dt < 2; ++dt)
        for (int grp = 0; grp < 4; ++grp) {       // 16-key group (i, k2) of the tile
          std::vector<int> addr(64);
          for (int l = 0; l < 64; ++l) {
            const int hi = l >> 5, i16 = l & 15, g1 = (l >> 4) & 1;
            const int key = 8 * r + 4 * hi + (i16 >> 2);
            const int chunk = dt * 4 + 2 * g1 + ((i16 & 3) >> 1);
            addr[l] = key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4) + (i16 & 1) * 8 + grp * 2048;
          }
          auto out = run(addr, img);
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
              const int hi = l >> 5, l31 = l & 31;
              const int want = (grp * 16 + 8 * r + 4 * hi + j) * 64 + dt * 32 + l31;
              if (out[l * 4 + j] != want) {
                if (ok) printf("  first mismatch: r %d dt %d group %d lane %d j %d: got key %d d %d, want key %d d %d\n", r, dt, grp, l, j,
                               out[l * 4 + j] / 64, out[l * 4 + j] % 64, want / 64, want % 64);
                ok = false;
              }
            }
        }
    printf("test 3 (attn_fwd_kernel7's addresses: lane (l31, hi) gets V[16 g + 8 r + 4 hi + j][32 dt + l31]): %s\n", ok ? "PASS" : "FAIL");
    if (!ok) ++failed;
  }
  return failed;
}
