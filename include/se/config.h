/*
 * Configuration -- the reference's application configuration struct, field for field
 *   se_denseslam/include/se/config.h:39-214
 * so that an application that fills it (se_apps/include/default_parameters.h:200-260 does, from the command line)
 * compiles unchanged against this build.  Same names, types and order; the Eigen types are the real ones when
 * <Eigen/Dense> is installed and the PODs of se/DenseSLAMSystem.h otherwise.  Three fields are appended for this build
 * (device 0, automatic brick layout, the streaming schedule that changes no observable result).
 */
#ifndef SE_HIP_CONFIG_H
#define SE_HIP_CONFIG_H

#include <string>
#include <vector>

#include "eigen_pods.h"

struct Configuration {
  int compute_size_ratio;              /* input frame size / computation size: 1, 2, 4 or 8 (default 1) */
  int tracking_rate;                   /* default 1 */
  int integration_rate;                /* default 2 */
  int rendering_rate;                  /* default 4 */
  Eigen::Vector3i volume_resolution;   /* default (256, 256, 256) */
  Eigen::Vector3f volume_size;         /* metres, default (2, 2, 2) */
  int voxel_block_size;
  Eigen::Vector3f initial_pos_factor;  /* default (0.5, 0.5, 0) */
  std::vector<int> pyramid;            /* default (10, 5, 4) */
  std::string dump_volume_file;
  std::string input_file;
  std::string log_file;
  std::string groundtruth_file;        /* "... tx ty tz qx qy qz qw" per line */
  Eigen::Matrix4f gt_transform;
  Eigen::Vector4f camera;              /* fx, fy, cx, cy */
  bool camera_overrided;
  float mu;                            /* TSDF truncation bound, default 0.1 */
  int fps;
  bool blocking_read;
  float icp_threshold;                 /* default 1e-5 */
  bool no_gui;
  bool render_volume_fullsize;
  bool bilateralFilter;
  bool colouredVoxels;                 /* unused in the reference */
  bool multiResolution;                /* unused in the reference */
  bool bayesian;                       /* unused in the reference */

  /* ---- additions of this build (not in the reference) */
  int hip_device = 0;                  /* HIP device ordinal of the map */
  long long hip_max_blocks = 0;        /* > 0: pooled bricks with this capacity instead of the dense brick grid */
  bool hip_streaming = true;           /* raycasting() may be launched together with the next integration()'s allocation scan (se_hip_set_streaming);
                                          nothing an application can observe through DenseSLAMSystem changes */
  bool hip_pinned_input = false;       /* preprocessing() reads an input image that lies in page-locked memory (DenseSLAMSystem::allocateInput) where it is,
                                          instead of copying it first: the buffer must then stay unmodified until the frame has been integrated
                                          (se_hip_set_pinned_input) */
};

#endif /* SE_HIP_CONFIG_H */
