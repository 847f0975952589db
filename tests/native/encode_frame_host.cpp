// A host with no Python in it: reads a frame + segment label maps from a file, encodes it through rhccq_encode_frame (include/rhccq.h), writes
// the palette and the index map.  tests/test_gpu_native_host.py builds it (hipcc, host code only), runs it as a child process and compares
// the output with FrameEncoder.encode on the same frame -- what SURVEY 8b means by a boundary a non-Python host can bind.
//   file in : int32 H, W, n_classes; per class int32 n_seg, n_region, quality, seg_region[n_seg], region_bbox[n_region * 4];
//             uint8 rgb[H * W * 3]; per class int32 labels[H * W]
//   file out: int32 n_colours, index_bytes, shape[2], top_left[2]; uint8 palette[n_colours * 3]; indices (index_bytes * H * W bytes)
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rhccq.h"

#define CHECK_HIP(e)                                                                  \
  do {                                                                                \
    hipError_t err_ = (e);                                                            \
    if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_)); return 2; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s frame.bin out.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 1;
  const int32_t H = hdr[0], W = hdr[1], n_classes = hdr[2];
  std::vector<std::vector<int32_t>> seg_region(n_classes), region_bbox(n_classes);
  std::vector<rhccq_class_desc> cls(n_classes);
  for (int c = 0; c < n_classes; ++c) {
    int32_t h3[3];
    if (fread(h3, 4, 3, f) != 3) return 1;
    seg_region[c].resize(h3[0]);
    region_bbox[c].resize((size_t)h3[1] * 4);
    if (fread(seg_region[c].data(), 4, h3[0], f) != (size_t)h3[0]) return 1;
    if (fread(region_bbox[c].data(), 4, (size_t)h3[1] * 4, f) != (size_t)h3[1] * 4) return 1;
    cls[c].n_seg = h3[0]; cls[c].n_region = h3[1]; cls[c].quality = h3[2]; cls[c].reserved = 0;
    cls[c].seg_region = seg_region[c].data(); cls[c].region_bbox = region_bbox[c].data();
  }
  const size_t n_px = (size_t)H * W;
  std::vector<uint8_t> rgb(n_px * 3);
  if (fread(rgb.data(), 1, rgb.size(), f) != rgb.size()) return 1;
  uint8_t* d_rgb;
  CHECK_HIP(hipMalloc((void**)&d_rgb, rgb.size()));
  CHECK_HIP(hipMemcpy(d_rgb, rgb.data(), rgb.size(), hipMemcpyHostToDevice));
  std::vector<int32_t> lab(n_px);
  for (int c = 0; c < n_classes; ++c) {
    if (fread(lab.data(), 4, n_px, f) != n_px) return 1;
    int32_t* d;
    CHECK_HIP(hipMalloc((void**)&d, n_px * 4));
    CHECK_HIP(hipMemcpy(d, lab.data(), n_px * 4, hipMemcpyHostToDevice));
    cls[c].labels = d;
  }
  fclose(f);
  rhccq_ctx* ctx = nullptr;
  if (rhccq_ctx_create(0, nullptr, &ctx)) { fprintf(stderr, "rhccq_ctx_create failed\n"); return 2; }
  void* d_idx;
  CHECK_HIP(hipMalloc(&d_idx, n_px * 4));
  std::vector<uint8_t> palette((size_t)3 << 16);
  rhccq_frame_result res;
  int rc = 0;
  for (int rep = 0; rep < 2; ++rep) {                     // twice: the second call runs on warm lanes and reused arenas
    rc = rhccq_encode_frame(ctx, d_rgb, H, W, cls.data(), n_classes, palette.data(), 1 << 16, d_idx, nullptr, &res);
    if (rc) { fprintf(stderr, "rhccq_encode_frame: %d (%s)\n", rc, rhccq_last_error(ctx)); return 3; }
  }
  std::vector<uint8_t> idx(n_px * (size_t)res.index_bytes);
  CHECK_HIP(hipMemcpy(idx.data(), d_idx, idx.size(), hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 1; }
  const int32_t oh[6] = {res.n_colours, res.index_bytes, res.shape[0], res.shape[1], res.top_left[0], res.top_left[1]};
  fwrite(oh, 4, 6, o);
  fwrite(palette.data(), 1, (size_t)res.n_colours * 3, o);
  fwrite(idx.data(), 1, idx.size(), o);
  fclose(o);
  printf("encoded %dx%d: %d colours, %d-byte indices, %.1f ms (levels 1-2 %.1f, level 3 %.1f)\n", W, H, res.n_colours, res.index_bytes, res.ms[6], res.ms[2],
         res.ms[3]);
  rhccq_ctx_destroy(ctx);
  return 0;
}
