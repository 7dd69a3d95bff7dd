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
#include <cstring>
#include <string>
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
  Configuration config;                          // the reference's struct, filled the way default_parameters.h:200-260 does
  config.compute_size_ratio = 1; config.tracking_rate = 1; config.integration_rate = 1; config.rendering_rate = 4;
  config.volume_resolution = Eigen::Vector3i(res, res, res); config.volume_size = Eigen::Vector3f(dim, dim, dim);
  config.initial_pos_factor = Eigen::Vector3f(0.f, 0.f, 0.f); config.pyramid = pyramid;
  config.dump_volume_file = ""; config.input_file = argv[1]; config.log_file = ""; config.groundtruth_file = argv[2];
  config.gt_transform = Eigen::Matrix4f::Identity(); config.camera = k; config.camera_overrided = argc >= 11;
  config.mu = mu; config.fps = 0; config.blocking_read = false; config.icp_threshold = 1e-5f; config.no_gui = true;
  config.render_volume_fullsize = false; config.bilateralFilter = false;
  config.colouredVoxels = false; config.multiResolution = false; config.bayesian = false;
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
  synchroniseDevices();                          // DenseSLAMSystem.h:418 (argument-less, as declared there)
  std::vector<float> vertex, normal;
  pipeline.getVertexNormal(vertex, normal);
  MapSnapshot map;
  pipeline.getMap(map);
  // se_apps/src/benchmark.cpp:179-187: getMap() -> save(), dump_volume()
  std::shared_ptr<se::Octree<FieldType> > map_ptr;
  pipeline.getMap(map_ptr);
  const std::string base(argv[6]);
  map_ptr->save(base + ".octree");
  pipeline.saveMap(base + ".devmap");
  pipeline.dump_volume(config.dump_volume_file);
  int fetch_bad = 0, coarse_checked = 0;
  for (auto& b : map_ptr->getBlockBuffer()) {
    const int* c = b->coordinates();
    if (map_ptr->fetch(c[0], c[1], c[2]) != b.get()) ++fetch_bad;
    if (map_ptr->fetch_octant(c[0], c[1], c[2], 64) != b.get()) ++fetch_bad;
    const auto v = map_ptr->get(c[0] + 3, c[1] + 4, c[2] + 5);
    const auto w = b->data(c[0] + 3, c[1] + 4, c[2] + 5);
    if (v.x != w.x || v.y != w.y || map_ptr->get_fine(c[0] + 3, c[1] + 4, c[2] + 5).x != w.x) ++fetch_bad;
  }
  for (auto& n : map_ptr->getNodesBuffer())   // Octree::get on unallocated space returns the parent's value_[child]
    for (int i = 0; i < 8; ++i)
      if (!n->child(i)) {
        const int h = (int)n->side_ / 2;
        int x = 0, y = 0, z = 0;
        const unsigned long long morton = n->code_ & ~0x1FFull;   // key = Morton code | level
        for (int b = 0; b < 21; ++b) { x |= (int)((morton >> (3 * b)) & 1ull) << b; y |= (int)((morton >> (3 * b + 1)) & 1ull) << b; z |= (int)((morton >> (3 * b + 2)) & 1ull) << b; }
        const auto v = map_ptr->get(x + ((i & 1) ? h : 0), y + ((i & 2) ? h : 0), z + ((i & 4) ? h : 0));
        if (v.x != n->value_[i].x) ++fetch_bad;
        ++coarse_checked;
      }
  // a second pipeline restored from the dump raycasts the same images (Octree::load counterpart)
  DenseSLAMSystem second(Eigen::Vector2i(W, H), Eigen::Vector3i(res, res, res), Eigen::Vector3f(dim, dim, dim),
                         Eigen::Vector3f(0.f, 0.f, 0.f), pyramid, config);
  int reload_identical = 0;
  if (second.loadMap(base + ".devmap") && frame > 0) {
    second.setPose(pipeline.getPose());
    second.raycasting(k, mu, frame - 1);
    std::vector<float> v2, n2;
    second.getVertexNormal(v2, n2);
    reload_identical = (v2.size() == vertex.size() && std::memcmp(v2.data(), vertex.data(), v2.size() * 4) == 0 &&
                        std::memcmp(n2.data(), normal.data(), n2.size() * 4) == 0) ? 1 : 0;
  }
  double sx = 0;
  for (float v : map.x) sx += v;
  FILE* out = std::fopen(argv[6], "wb");
  const int32_t hdr[4] = {W, H, map.n_blocks, (int32_t)frame};
  std::fwrite(hdr, 4, 4, out);
  std::fwrite(vertex.data(), 4, vertex.size(), out);
  std::fwrite(normal.data(), 4, normal.size(), out);
  std::fclose(out);
  std::printf("frames %u raycast %d blocks %d nodes %d sum_x %.3f\n", frame, (int)raycast_ran, map.n_blocks, map.n_nodes, sx);
  std::printf("octree blocks %zu nodes %zu fetch_bad %d coarse_checked %d reload_identical %d\n", map_ptr->getBlockBuffer().size(),
              map_ptr->getNodesBuffer().size(), fetch_bad, coarse_checked, reload_identical);
  return 0;
}
