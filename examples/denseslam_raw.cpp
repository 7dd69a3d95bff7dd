// Minimal C++ application against the DenseSLAMSystem mirror: reads a SLAMBench ".raw" depth
// stream (layout of se_tools/scene2raw.cpp:170-176: per frame uint32 w,h + uint16 depth[w*h] +
// uint32 w,h + uchar3 rgb[w*h]) and a pose file (16 floats per frame, row-major camera->world),
// runs preprocessing -> setPose -> integration -> raycasting like se_apps/src/benchmark.cpp:115-177
// with ground-truth poses, and writes the last frame's vertex / normal maps plus a map summary.
//   usage: denseslam_raw <scene.raw> <poses.bin> <volume_res> <volume_dim> <mu> <out.bin> [fx fy cx cy]
#ifndef SE_FIELD_TYPE
#define SE_FIELD_TYPE SDF
#endif
#include <se/DenseSLAMSystem.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: %s scene.raw poses.bin res dim mu out.bin [fx fy cx cy]\n", argv[0]); return 2; }
  FILE* raw = std::fopen(argv[1], "rb");
  FILE* pf = std::fopen(argv[2], "rb");
  if (!raw || !pf) { std::fprintf(stderr, "cannot open inputs\n"); return 2; }
  const int res = std::atoi(argv[3]);
  const float dim = (float)std::atof(argv[4]), mu = (float)std::atof(argv[5]);
  uint32_t wh[2];
  if (std::fread(wh, 4, 2, raw) != 2) return 2;
  std::fseek(raw, 0, SEEK_SET);
  const int W = (int)wh[0], H = (int)wh[1];
  Eigen::Vector4f k(481.2f * W / 640.f, 480.f * W / 640.f, 320.f * W / 640.f, 240.f * W / 640.f);
  if (argc >= 11) k = Eigen::Vector4f((float)std::atof(argv[7]), (float)std::atof(argv[8]), (float)std::atof(argv[9]), (float)std::atof(argv[10]));
  std::vector<int> pyramid = {10, 5, 4};
  Configuration config;
  config.mu = mu;
  DenseSLAMSystem pipeline(Eigen::Vector2i(W, H), Eigen::Vector3i(res, res, res), Eigen::Vector3f(dim, dim, dim),
                           Eigen::Vector3f(0.f, 0.f, 0.f), pyramid, config);
  std::vector<unsigned short> depth((size_t)W * H);
  std::vector<unsigned char> rgb((size_t)W * H * 3);
  float pose_rm[16];
  unsigned frame = 0;
  bool raycast_ran = false;
  while (std::fread(wh, 4, 2, raw) == 2) {
    if (std::fread(depth.data(), 2, depth.size(), raw) != depth.size()) break;
    if (std::fread(wh, 4, 2, raw) != 2 || std::fread(rgb.data(), 1, rgb.size(), raw) != rgb.size()) break;
    if (std::fread(pose_rm, 4, 16, pf) != 16) break;
    Eigen::Matrix4f pose;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) pose(r, c) = pose_rm[r * 4 + c];
    pipeline.preprocessing(depth.data(), Eigen::Vector2i(W, H), false);
    pipeline.setPose(pose);                       // init position is the origin here
    pipeline.integration(k, 1, mu, frame);
    raycast_ran = pipeline.raycasting(k, mu, frame);
    ++frame;
  }
  synchroniseDevices(pipeline);
  std::vector<float> vertex, normal;
  pipeline.getVertexNormal(vertex, normal);
  MapSnapshot map;
  pipeline.getMap(map);
  double sx = 0;
  for (float v : map.x) sx += v;
  FILE* out = std::fopen(argv[6], "wb");
  const int32_t hdr[4] = {W, H, map.n_blocks, (int32_t)frame};
  std::fwrite(hdr, 4, 4, out);
  std::fwrite(vertex.data(), 4, vertex.size(), out);
  std::fwrite(normal.data(), 4, normal.size(), out);
  std::fclose(out);
  std::printf("frames %u raycast %d blocks %d nodes %d sum_x %.3f\n", frame, (int)raycast_ran, map.n_blocks, map.n_nodes, sx);
  return 0;
}
