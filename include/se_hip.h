/*
 * se_hip.h -- C ABI of the MI355X (gfx950) dense-fusion hot path.
 *
 * This is the drop-in boundary for supereight's per-frame path
 *     depth image -> voxel-block allocation -> TSDF / occupancy integration -> raycast
 * i.e. what se_denseslam's DenseSLAMSystem::integration() / ::raycasting()
 * (se_denseslam/src/DenseSLAMSystem.cpp:191-268) do on the host.  The reference has no FFI of
 * its own (one process, header-only C++); each entry point below cites the reference
 * interface it replaces.  POD arguments only: the same shared library serves the C++
 * `DenseSLAMSystem` mirror (include/se/DenseSLAMSystem.h, header-only), the ctypes binding
 * (supereight_amd/pipeline.py) and any other FFI.
 *
 * Conventions
 *   - every function returns an int status: >= 0 success (stage functions return 1 = "ran this
 *     frame", 0 = "gated off", like the reference's bool), < 0 = SE_HIP_E_* ; nothing throws,
 *     nothing calls exit(); se_hip_last_error() gives a message for the calling thread.
 *   - 4x4 matrices are 16 floats in COLUMN-MAJOR order, i.e. Eigen::Matrix4f::data().
 *   - k = (fx, fy, cx, cy) as Eigen::Vector4f k in the reference API.  A pose or k holding a NaN or an infinity (or fx, fy = 0) is refused with
 *     SE_HIP_E_INVALID by every stage call: the reference would fuse garbage; this library's parity argument is made for finite rays.
 *   - one handle <-> one caller thread at a time (the reference is not re-entrant either).
 *   - all work is enqueued on one HIP stream per handle; calls that return data to the host
 *     synchronise that stream, the others are asynchronous -- with one exception (the "host gate", dense unsharded
 *     handles): se_hip_alloc_scan / se_hip_integrate / se_hip_frame wait on the HOST until the
 *     raycast behind the previous integration sweep has started on the device (normally less than a frame period; after
 *     20 ms of wall clock the wait turns into hipStreamSynchronize of the handle's stream); a depth upload waits the same way for the raycast
 *     behind the sweep of three uploads ago (the slot it overwrites).  A caller that hands in its own
 *     stream (se_hip_set_stream) must therefore not hold that stream behind work it has not enqueued yet.  SE_HIP_HOST_GATE=0
 *     in the environment selects the event-ordered form, in which no stage call waits on the host.
 *   - SE_HIP_E_CAPACITY is sticky: once a pool, key list or brick segment has overflowed, every later stage call and
 *     se_hip_sync return it (the map / the replicas are no longer what the reference would hold) until se_hip_load_map
 *     re-initialises the map or the caller acknowledges it with se_hip_clear_overflow().
 */
#ifndef SE_HIP_H
#define SE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SE_FIELD_TYPE of the reference (se_denseslam/include/se/volume_traits.hpp:41-72;
 * se_denseslam/CMakeLists.txt:31-50 builds one library per field type) */
#define SE_HIP_FIELD_SDF 0
#define SE_HIP_FIELD_OFUSION 1

#define SE_HIP_OK 0
#define SE_HIP_E_INVALID (-1)   /* bad argument */
#define SE_HIP_E_DEVICE (-2)    /* HIP runtime error */
#define SE_HIP_E_CAPACITY (-3)  /* block / node / key-list pool exhausted */
#define SE_HIP_E_NOGPU (-4)     /* no usable gfx950 device */

typedef struct se_hip_pipeline se_hip_pipeline;

/* Replaces the state set up by DenseSLAMSystem's constructor
 * (se_denseslam/src/DenseSLAMSystem.cpp:65-126: computation_size_, volume_resolution_,
 * volume_dimension_, discrete_vol_ptr_->init(res, dim)). */
typedef struct se_hip_config {
  int32_t width;             /* computation_size_.x() */
  int32_t height;            /* computation_size_.y() */
  int32_t volume_resolution; /* voxels per side; power of two in [64, 4096] */
  float volume_dimension;    /* metres per side */
  int32_t field_type;        /* SE_HIP_FIELD_* */
  int32_t device;            /* HIP device ordinal */
  int64_t max_blocks;        /* capacity of the voxel-block pool; 0 = default (a dense brick grid while it costs <= 64 GiB and a third of the free device memory, else 24 (N/8)^2 pooled bricks) */
  int32_t row_begin;         /* image rows [row_begin,row_end) this handle alloc-scans and */
  int32_t row_end;           /*   raycasts (multi-GPU tile sharding); 0,0 = the whole image */
} se_hip_config;

int se_hip_create(const se_hip_config* cfg, se_hip_pipeline** out);
/* The multi-device form of the constructor (device_ids[]): one row-sharded replica per listed device,
 * replica i owning the i-th share of the image's 8-row tiles (cfg->device / row_begin / row_end are ignored).  The
 * per-frame key-list exchange between the replicas is the caller's (se_hip_alloc_exchange, or se_hip_new_keys_device +
 * se_hip_alloc_commit); one process per GPU with se_hip_create is the deployment this library is measured in. */
int se_hip_create_replicas(const se_hip_config* cfg, const int32_t* device_ids, int32_t n_devices, se_hip_pipeline** out_handles);
int se_hip_destroy(se_hip_pipeline* p);
const char* se_hip_last_error(void);
/* free function synchroniseDevices() is declared but never defined in the reference
 * (se_denseslam/include/se/DenseSLAMSystem.h:418); this is its body. */
int se_hip_sync(se_hip_pipeline* p);
/* Acknowledges a reported SE_HIP_E_CAPACITY (see the conventions above): synchronises, clears the device-side overflow flag and
 * returns the code that was pending (0 = none, 1 = block / node pool, 2 = key list, 3 = brick segment).  The map keeps whatever
 * it lost; the call only lets a caller that has dealt with that (e.g. by reloading the map) carry on. */
int se_hip_clear_overflow(se_hip_pipeline* p);
/* Use an existing hipStream_t (e.g. PyTorch's current stream) instead of the handle's own. */
int se_hip_set_stream(se_hip_pipeline* p, void* hip_stream);
/* The stream the allocation scan (se_hip_alloc_scan) is launched on when it overlaps the previous
 * frame's raycast.  The multi-GPU driver passes the stream its
 * all-gather of the key lists is ordered on, so that scan + exchange of frame f+1 hide behind the
 * raycast of frame f.  NULL = a stream owned by the handle (the default) -- NOT the legacy default stream: a caller whose collective runs on
 * stream 0 (PyTorch's default stream is handle 0) must create a real stream for it and pass that, or scan and collective are unordered. */
int se_hip_set_scan_stream(se_hip_pipeline* p, void* hip_stream);
/* 1 if the key list of se_hip_alloc_scan is produced on the scan stream (overlap on: the default), 0 if on the main
 * stream like every other stage -- in which case work ordered with the scan (an all-gather of its list) belongs on the main
 * stream.  With overlap on, a handle that is row-sharded, writes its list into a caller buffer (se_hip_set_new_keys_buffer)
 * or has an exchange set (se_hip_set_exchange) ALWAYS scans on the scan stream; only a plain single handle whose main
 * stream is idle at the call (a caller that synchronises every frame) gets the scan straight onto the main stream, and
 * nobody else consumes its list.  se_hip_alloc_exchange / se_hip_alloc_commit follow the stream the last scan ran on. */
int se_hip_scan_overlaps(se_hip_pipeline* p);
/* ---- streaming callers: the one-queue schedule (off by default).
 * The reference's loop is integration(f); raycasting(f); integration(f+1); ... (se_apps/src/benchmark.cpp:148-167).  A caller that streams frames
 * without looking at each frame's vertex_ / normal_ can have raycasting(f) and the allocation scan of integration(f+1) run as ONE launch on the
 * handle's stream: with se_hip_set_streaming(p, 1), se_hip_frame and se_hip_raycast_deferred do not enqueue the raycast of a frame but hold it back
 * until the next se_hip_integrate / se_hip_frame / se_hip_alloc_scan, which launches it together with that frame's scan.  ANY other entry point of
 * this header (se_hip_sync, the image getters, se_hip_raycast, se_hip_track, the render calls, ...; not: the depth uploads, se_hip_filter_depth,
 * se_hip_set_new_keys_buffer, se_hip_enable_timing, se_hip_get_launch_counts) launches an outstanding raycast first, so whatever is read THROUGH the API is what the eager
 * schedule gives.  Reading past the API is what the mode cannot make safe: a caller kernel ordered on the handle's stream behind se_hip_frame(f)
 * would find frame f-1's images.  Therefore
 *   - the mode is opt-in (off: se_hip_frame == se_hip_set_depth_device + se_hip_integrate + se_hip_raycast, all enqueued when it returns);
 *   - se_hip_vertex_normal_device on a streaming handle WITHOUT an image ring switches deferral off for good (sticky): raw pointers and deferral
 *     never coexist;
 *   - with an image ring (below) the contract is explicit: slot (f % slots) holds frame f's images once the NEXT se_hip_frame / se_hip_integrate
 *     call (or any flushing call) has returned and the handle's stream has reached that point;
 *   - a caller that turns out to look at every frame -- the reference's loop with tracking on: se_hip_track(f+1) needs raycasting(f)'s images -- gains
 *     nothing from holding raycasts back (they start later and fuse with no scan).  After two held-back raycasts in a row that some other call had
 *     to launch, with no fused launch in between, the handle launches raycasts eagerly again (r06; se_hip_set_streaming(p, 1) re-arms deferral).
 * Handles whose raycast fits the chip in one round of workgroups (640x480: 2 400 of 2 560) fuse; others (statistics on, sharded sweep, larger
 * images) keep the eager two-queue schedule whatever the flag says.  Row-sharded replicas and handles with a caller key buffer or an exchange fuse
 * too: the scan half of the launch writes the caller's list on the MAIN stream, and se_hip_alloc_exchange / se_hip_alloc_commit follow it there
 * (se_hip_scan_overlaps still answers for the scans that do not ride in a raycast's launch: give such a handle the main stream as its scan
 * stream, se_hip_set_scan_stream, so that both kinds are ordered with the caller's collective -- supereight_amd/multi_gpu.py does).
 * se_hip_set_streaming returns 1 if the handle will fuse, 0 if not (or off); se_hip_frame_is_fused reports the same without changing anything. */
int se_hip_set_streaming(se_hip_pipeline* p, int32_t on);
int se_hip_frame_is_fused(se_hip_pipeline* p);
/* vertex_ / normal_ into a caller-owned device ring instead of the handle's own images: the raycast of frame f -- eager, deferred or fused --
 * writes slot f % slots, a slot = [vertex: width*height*3 floats][normal: width*height*3 floats] (se_hip_image_tile_bytes(p, height) bytes), and
 * that slot IS vertex_ / normal_ for every later consumer (se_hip_track, the render calls, the getters) until the next raycast.  This is how a
 * streaming caller keeps every frame's images (tests/test_gpu_stress_parity.py compares each slot of a fused stream with the oracle).
 * NULL, 0 restores the handle's own images (the current images are copied back; synchronises). */
int se_hip_set_image_ring(se_hip_pipeline* p, float* device_ring, int32_t slots);

/* ---- input: float_depth_ (se::Image<float>, metres, row-major x + y*w), produced by
 * preprocessing() in the reference (DenseSLAMSystem.cpp:128-141).
 * Host images (r06): the call copies the caller's buffer into a ring of three pinned host buffers and returns -- like the reference's synchronous
 * preprocessing(), the caller may reuse its buffer at once -- and enqueues NOTHING: the first kernel of the frame that needs float_depth_ (the allocation
 * scan, one thread per pixel; or a one-kernel conversion in front of se_hip_track / se_hip_render_depth / a row-sharded scan) reads the pinned image over
 * PCIe and writes the device image on its way.  No DMA packet, no second queue, no wait for the previous frame: on a streaming handle the input of frame
 * f+1 crosses PCIe inside the launch that raycasts frame f.  The call waits (on the host) only if all three slots are still in flight, i.e. if the caller
 * is three uploads ahead of the device. */
int se_hip_upload_depth(se_hip_pipeline* p, const float* host_depth_m);
/* mm2metersKernel (se_denseslam/src/preprocessing.cpp:161-188) applied where the image is first read on the device:
 * uint16 millimetres of size (in_w, in_h), an integer multiple of the computation size ("Invalid ratio." = SE_HIP_E_INVALID). */
int se_hip_upload_depth_mm(se_hip_pipeline* p, const uint16_t* host_depth_mm, int32_t in_w, int32_t in_h);
/* Caller-pinned input (opt-in, r06): with se_hip_set_pinned_input(p, 1) an image handed to the two calls above that lies in page-locked host memory
 * (se_hip_host_alloc below, hipHostMalloc, hipHostRegister) is NOT copied: the frame's first kernel reads it over PCIe where the caller keeps it -- a
 * reader that decodes its frames straight into such a buffer (the reference's loop reads a frame per iteration, se_apps/src/benchmark.cpp:115-133) saves
 * the 16 us copy of a 640x480 image per frame.  The price is the reference's synchronous contract: the buffer must stay unmodified until the frame's
 * integration has run on the device -- i.e. until a call that waits for it has returned (se_hip_sync, an image download, se_hip_track of the next
 * frame) or, for a streaming caller, until three further uploads have been accepted (the handle's own ring discipline).  Pageable images are copied as
 * before, whatever the flag says.  se_hip_host_alloc / se_hip_host_free: page-locked host memory without a HIP dependency in the caller (NULL on failure). */
int se_hip_set_pinned_input(se_hip_pipeline* p, int32_t on);
void* se_hip_host_alloc(size_t bytes);
void se_hip_host_free(void* host);
/* Zero-copy: integrate from a depth image already resident in HBM (width*height floats).  The buffer is read by the
 * allocation scan and by the integration sweep of the frame: it must stay untouched until that sweep has finished
 * (se_hip_sync, or work ordered behind se_hip_integrate on the handle's stream).  It must also be COMPLETE when the handle's streams get to
 * it: a producer kernel on another stream is not ordered with them -- hand the handle the producer's stream (se_hip_set_stream) or wait for it. */
int se_hip_set_depth_device(se_hip_pipeline* p, const float* device_depth_m);

/* ---- bool DenseSLAMSystem::integration(const Vector4f& k, unsigned integration_rate, float mu,
 *      unsigned frame)  (DenseSLAMSystem.h:193, DenseSLAMSystem.cpp:206-268); `pose` is the
 *      member pose_ (camera -> world). */
int se_hip_integrate(se_hip_pipeline* p, const float pose[16], const float k[4], uint32_t integration_rate, float mu,
                     uint32_t frame);
/* The same stage split for multi-GPU runs, so that the caller can put the RCCL allgather of the
 * per-rank new-block key lists between the allocation scan and the sweep:
 *   se_hip_alloc_scan     = buildAllocationList / buildOctantList + Octree::allocate for the keys
 *                           found in this handle's image rows (kfusion/alloc_impl.hpp:54-118,
 *                           bfusion/alloc_impl.hpp:56-129, se_core/include/se/octree.hpp:792-856)
 *   se_hip_new_keys_device= the list this scan produced: uint64[0] = count, uint64[1..count] =
 *                           keys in the reference's key format (octant_ops.hpp:49-53)
 *   se_hip_alloc_commit   = Octree::allocate for `nlists` such lists gathered from other ranks
 *                           (device memory, list i at device_lists + i*stride_words)
 *   se_hip_integrate_sweep= projective_map (se_core/include/se/functors/projective_functor.hpp:139-176) */
int se_hip_alloc_scan(se_hip_pipeline* p, const float pose[16], const float k[4], uint32_t integration_rate, float mu,
                      uint32_t frame);
int se_hip_new_keys_device(se_hip_pipeline* p, uint64_t** device_list, int64_t* capacity_words);
/* Make the scan write its list into caller-owned device memory (e.g. the send buffer of the RCCL
 * allgather); capacity_words includes the count word.  NULL restores the internal buffer. */
int se_hip_set_new_keys_buffer(se_hip_pipeline* p, uint64_t* device_list, int64_t capacity_words);
/* The lists must have been produced by work ordered on the stream the allocation scan runs on: the scan stream
 * (se_hip_set_scan_stream) when se_hip_scan_overlaps() is 1 -- the commit kernel is launched there, behind the scan and
 * the caller's all-gather, beside the previous frame's raycast -- and the main stream otherwise.  A new-key list that
 * overflowed (more keys than its capacity) makes the next stage call fail with SE_HIP_E_CAPACITY: the replicas would
 * diverge otherwise. */
int se_hip_alloc_commit(se_hip_pipeline* p, const uint64_t* device_lists, int32_t nlists, int64_t stride_words);
/* Multi-GPU exchange without a host framework in the per-frame path: `nccl_comm` is the ncclComm_t of the caller's
 * communicator (RCCL), `nccl_all_gather` the address of ncclAllGather in the RCCL the process has loaded (the library
 * itself does not link RCCL).  se_hip_alloc_exchange all-gathers the first `words` words of the scan's key list
 * (se_hip_set_new_keys_buffer / the internal list) into recv_device (world * words) on the scan stream and then does
 * se_hip_alloc_commit on the gathered lists.  NULL, NULL switches it off. */
int se_hip_set_exchange(se_hip_pipeline* p, void* nccl_comm, void* nccl_all_gather, int32_t world);
int se_hip_alloc_exchange(se_hip_pipeline* p, uint64_t* recv_device, int64_t words);
int se_hip_integrate_sweep(se_hip_pipeline* p, const float pose[16], const float k[4], uint32_t integration_rate,
                           float mu, uint32_t frame);
/* Sharded sweep -- SURVEY 8(e) option 4, the alternative to the replicated sweep of projective_functor::apply
 * (projective_functor.hpp:139-160) above; measured and priced in DESIGN.md section 7, off by default.  Replica `rank` of
 * `world` updates only the blocks it owns (owner = (bx + by + bz) mod world, block units) and packs, for each of them, its
 * position, its new active flag and -- if any voxel was in view -- its 512 voxels into the caller's send segment of
 * se_hip_sweep_shard_bytes(cap_bricks) bytes (16-byte aligned), cap_bricks a multiple of 64: [u64 counts x 64][u32 records x cap][float vx
 * x 512 x cap][float vy x 512 x cap], counter c over the records [c * cap / 64, (c + 1) * cap / 64).  After the caller's all-gather of the segments (rank order), se_hip_apply_bricks writes the other replicas'
 * records into this replica's map on the main stream; se_hip_brick_exchange issues that all-gather itself
 * (se_hip_set_exchange) on the main stream and then applies.  A segment that overflows makes the next stage call fail with
 * SE_HIP_E_CAPACITY.  OFusion's node values stay replicated (every replica updates every node).  world <= 1 switches it off. */
size_t se_hip_sweep_shard_bytes(size_t cap_bricks);
int se_hip_set_sweep_shard(se_hip_pipeline* p, int32_t rank, int32_t world, void* send_device, size_t cap_bricks);
int se_hip_apply_bricks(se_hip_pipeline* p, const void* recv_device, int32_t world);
int se_hip_brick_exchange(se_hip_pipeline* p, void* recv_device);

/* Full vertex_ / normal_ images on every rank of a row-sharded run: se_hip_raycast of a sharded handle fills
 * only the handle's own image rows, but their consumer -- tracking(), DenseSLAMSystem.cpp:175-177, and the render*() methods --
 * reads the whole images.  A tile = the rows [row_begin, row_end) of a rank, padded to max_rows rows (the largest share of
 * the partition), vertex rows first, then normal rows: se_hip_image_tile_bytes(p, max_rows) bytes.
 *   se_hip_pack_image_tile    copies the handle's own rows of both images into send_device (one tile);
 *   se_hip_apply_image_tiles  writes the other ranks' tiles (recv_device = world tiles in rank order, as an all-gather of the
 *                             packed tiles delivers them; row_begin / row_end = the partition) into this handle's images;
 *   se_hip_gather_images      pack + ncclAllGather on the communicator of se_hip_set_exchange + apply.
 * Everything is enqueued on the main stream, behind the raycast. */
size_t se_hip_image_tile_bytes(se_hip_pipeline* p, int32_t max_rows);
int se_hip_pack_image_tile(se_hip_pipeline* p, void* send_device, int32_t max_rows);
int se_hip_apply_image_tiles(se_hip_pipeline* p, const void* recv_device, int32_t world, int32_t max_rows, const int32_t* row_begin, const int32_t* row_end);
int se_hip_gather_images(se_hip_pipeline* p, void* send_device, void* recv_device, int32_t max_rows, const int32_t* row_begin, const int32_t* row_end);

/* One frame of the loop of se_apps/src/benchmark.cpp:148-167 in one call: hand-over of a device-resident float_depth_
 * (NULL = keep the current depth image), then integration(), then raycasting() with the same pose -- exactly
 * se_hip_set_depth_device + se_hip_integrate + se_hip_raycast (se_hip_raycast_deferred on a streaming handle, see se_hip_set_streaming).
 * Returns bit 0 = integration ran, bit 1 = raycasting ran (or is held back).  The depth image handed over must stay valid until the next
 * call (the scan and the sweep read it asynchronously). */
int se_hip_frame(se_hip_pipeline* p, const float* device_depth_m, const float pose[16], const float k[4], uint32_t integration_rate,
                 float mu, uint32_t frame);

/* ---- bool DenseSLAMSystem::raycasting(const Vector4f& k, float mu, unsigned frame)
 *      (DenseSLAMSystem.h:212, DenseSLAMSystem.cpp:191-204) -> vertex_, normal_ */
int se_hip_raycast(se_hip_pipeline* p, const float pose[16], const float k[4], float mu, uint32_t frame);
/* raycasting() of a streaming caller (se_hip_set_streaming): same gate, same result, but the launch is held back until the next
 * se_hip_integrate / se_hip_frame (one launch with that frame's allocation scan) or any flushing call; on a handle that does not fuse it IS
 * se_hip_raycast.  include/se/DenseSLAMSystem.h's raycasting() calls this one. */
int se_hip_raycast_deferred(se_hip_pipeline* p, const float pose[16], const float k[4], float mu, uint32_t frame);
/* vertex_ / normal_ : se::Image<Eigen::Vector3f>, packed 12 bytes per pixel, world frame */
int se_hip_download_vertex_normal(se_hip_pipeline* p, float* host_vertex_xyz, float* host_normal_xyz);
/* The device images themselves (zero-copy consumers).  They are current when the call returns (an outstanding deferred raycast is launched first)
 * and are rewritten by the next raycast on the handle's stream.  On a streaming handle without an image ring the call ends deferral for good
 * (see se_hip_set_streaming); with a ring it returns the slot of the last raycast. */
int se_hip_vertex_normal_device(se_hip_pipeline* p, float** device_vertex_xyz, float** device_normal_xyz);

/* ---- "next" row f-2: bool DenseSLAMSystem::tracking(const Vector4f& k, float icp_threshold,
 *      unsigned tracking_rate, unsigned frame)  (DenseSLAMSystem.h:173, DenseSLAMSystem.cpp:143-189):
 *      half-sample pyramid of the current depth image, depth2vertex / vertex2normal per level, ICP against vertex_ /
 *      normal_ of the last se_hip_raycast, checkPoseKernel.  The ICP loop is device-resident: one launch per iteration
 *      (k_icp_iter: the previous iteration's final sums + updatePoseKernel -- 6x6 Cholesky solve, SE3 exponential, pose update,
 *      convergence test -- as a prologue, then trackKernel + reduceKernel's partial sums) and a last launch that is k_icp_finish (the
 *      last iteration's sums and update, checkPoseKernel, the host record) in its first workgroup and tracking_result_ in the others;
 *      the pyramid is one launch (copy + two half-samplings), vertices + normals of all levels another.  The call waits on the host
 *      once for the result; while it enqueues a level's iterations it stays two launches ahead of the device and stops enqueuing a
 *      level that has converged (the launches left out would have returned at once; SE_HIP_ICP_LOOKAHEAD=0: all up front).
 *      A row-sharded handle must have the peers' rows of vertex_ / normal_ (se_hip_gather_images or se_hip_apply_image_tiles
 *      after the raycast): SE_HIP_E_INVALID otherwise.  se_hip_download_track: `result` of every pixel and error / J of the
 *      accepted ones are the reference's; its rejected pixels keep leftovers of earlier iterations there, zeros here.
 *      pose_inout = pose_ (updated in place; restored if the check fails); pyramid = iterations per
 *      level, finest first (default {10, 5, 4}).  Returns 1 = tracked, 0 = gated off or rejected. */
int se_hip_track(se_hip_pipeline* p, const float k[4], float icp_threshold, uint32_t tracking_rate, uint32_t frame,
                 const int32_t* pyramid, int32_t n_levels, float pose_inout[16]);
/* One frame of the reference's loop with tracking on (se_apps/src/benchmark.cpp:115-150) in one call:
 *   float_depth_ = device_depth_m (NULL: keep);  tracked = tracking();  if (tracked || frame <= 3) integration();  raycasting();
 * pose_inout: pose_ (in), pose_ after tracking (out).  Returns bit 0: integration ran, bit 1: raycasting ran, bit 2: tracked;
 * < 0 on error.  Same results as se_hip_set_depth_device + se_hip_track + se_hip_integrate + se_hip_raycast. */
int se_hip_frame_tracked(se_hip_pipeline* p, const float* device_depth_m, const float k[4], float icp_threshold, uint32_t tracking_rate,
                         const int32_t* pyramid, int32_t n_levels, float pose_inout[16], uint32_t integration_rate, float mu, uint32_t frame);
/* preprocessing(..., filterInput) (DenseSLAMSystem.cpp:128-141): when on, se_hip_track works on
 * bilateralFilterKernel(float_depth_) (preprocessing.cpp:41-89, gaussian_ of DenseSLAMSystem.cpp:111-118)
 * instead of float_depth_ itself; integration always uses the unfiltered image, as in the reference. */
int se_hip_filter_depth(se_hip_pipeline* p, int32_t on);
/* scaled_depth_[level] as built by the last se_hip_track ((width >> level) x (height >> level) floats). */
int se_hip_download_scaled_depth(se_hip_pipeline* p, int32_t level, float* host_out);
/* tracking_result_ (TrackData {int result; float error; float J[6];} per pixel, commons.h:249-253) and
 * row 0 of reduction_output_ (32 floats) of the last ICP iteration; iterations run in the last call. */
int se_hip_download_track(se_hip_pipeline* p, void* host_trackdata, float host_reduce32[32], int32_t* iterations);

/* ---- "next" row f-3: the render*() methods (DenseSLAMSystem.h:241-286, DenseSLAMSystem.cpp:274-300;
 *      kernels se_denseslam/src/rendering.cpp:111-283).  Output: width*height RGBW bytes (host).
 *      se_hip_render_volume: view_pose = *viewPose_ (re-raycasts with far = 2*farPlane when it is not
 *      approximately raycast_pose_, else shades vertex_/normal_); light = its translation, ambient 0.1;
 *      mu = the constructor's config.mu; returns 1 = rendered, 0 = gated off (frame % rate != 0). */
int se_hip_render_volume(se_hip_pipeline* p, uint8_t* host_rgbw, const float view_pose[16], const float k[4], float mu, float largestep,
                         uint32_t frame, uint32_t raycast_rendering_rate);
int se_hip_render_depth(se_hip_pipeline* p, uint8_t* host_rgbw);
int se_hip_render_track(se_hip_pipeline* p, uint8_t* host_rgbw);

/* ---- map read-back: what getMap() exposes as a host se::Octree
 *      (DenseSLAMSystem.h:295; se_core/include/se/octree.hpp:898-914 save layout). */
int se_hip_counts(se_hip_pipeline* p, int32_t* n_blocks, int32_t* n_nodes);
/* What the map costs on the device (no reference counterpart: MemoryPool grows on the host heap, se_core/include/se/utils/memory_pool.hpp:64-95):
 * out[0] = 1 dense brick grid / 0 pooled bricks, out[1] = brick slots, out[2] = bytes of the voxel bricks, out[3] = bytes of everything the handle
 * holds on the device (bricks, index pyramid, bitmaps, lists, key buffers, images, input ring).  Does not touch the device. */
int se_hip_memory_info(se_hip_pipeline* p, int64_t out[4]);
/* blocks sorted by key: coords[n][3] (min corner, voxels), x[n][512], y[n][512] (voxel index
 * x + 8y + 64z, se_core/include/se/node.hpp:139-144), active[n] */
int se_hip_download_blocks(se_hip_pipeline* p, int32_t* coords, float* x, float* y, uint8_t* active);
/* internal nodes sorted by key: code[n] (key = code|level), side[n], x[n][8], y[n][8] (value_[8]) */
int se_hip_download_nodes(se_hip_pipeline* p, uint64_t* code, uint32_t* side, float* x, float* y);

/* ---- "next" row f-4: Octree::save (se_core/include/se/octree.hpp:898-914, io/se_serialise.hpp:54-86),
 *      written straight from the device map in the reference's byte layout:
 *        int32 size, float dim, uint64 n_nodes, n_nodes x {uint64 code, int32 side, value_[8]},
 *        uint64 n_blocks, n_blocks x {uint64 code, int32 coords[3], voxel_block_[512]}
 *      with value_type = {float x, float y} (SDF) or {float x, 4 pad bytes, double y} (OFusion).
 *      Nodes and blocks are written sorted by key (the reference writes its pool order, which is
 *      nondeterministic under OpenMP). */
int se_hip_save_map(se_hip_pipeline* p, const char* filename);
/* Octree::load (se_core/include/se/octree.hpp:917-950): re-initialises the device map and restores it from a file in
 * the layout above (written by se_hip_save_map or by the reference's Octree::save) for the same size / dim / field
 * type.  The reference's load() reads `dim` as an int and restores one voxel per block (octree.hpp:921-924, 945-946);
 * this one reads the float and restores all 512.  Blocks come back active, as Octree::insert leaves them. */
int se_hip_load_map(se_hip_pipeline* p, const char* filename);

/* ---- map export, second half: marching cubes over the allocated blocks
 * DenseSLAMSystem::dump_mesh (DenseSLAMSystem.cpp:302-322) = se::algorithms::marching_cube
 * (se_core/include/se/algorithms/meshing.hpp:161-208) with inside(v) = v.x < 0, select(v) = v.x, then writeVtkMesh
 * (se_denseslam/include/se/commons.h:325-390).  Vertices (which cell edges carry one, and where) are the
 * reference's, and so is the split of a cell's polygon into triangles: include/se_mc_table.h is the standard marching-cubes
 * table, content-identical to the reference's edge_tables.h:66 (tests/test_mc_table_reference.py).  A triangle is 9 floats (3 vertices, metres); their order is unspecified, as in the
 * reference (OpenMP completion order there). */
int se_hip_mesh_count(se_hip_pipeline* p, int64_t* n_triangles);
int se_hip_mesh_download(se_hip_pipeline* p, float* host_triangles, int64_t capacity_triangles, int64_t* n_written);
/* dump_mesh(filename): ASCII VTK polydata in writeVtkMesh's format, triangles sorted for reproducibility */
int se_hip_dump_mesh(se_hip_pipeline* p, const char* filename);

/* ---- measurement (replaces TICK()/TOCK() + PerfStats, se_shared/timings.h:7-15) */
#define SE_HIP_K_ALLOC_SCAN 0
#define SE_HIP_K_ALLOC_COMMIT 1
#define SE_HIP_K_INTEGRATE 2 /* block sweep + node sweep, one launch */
#define SE_HIP_K_RAYCAST 3
#define SE_HIP_K_APPLY_BRICKS 4 /* sharded sweep: the other replicas' bricks written into this map */
#define SE_HIP_K_COUNT 5
/* HIP-event timing of every kernel launch on the handle's stream (off by default). */
int se_hip_enable_timing(se_hip_pipeline* p, int32_t on);
/* sum of launch durations [ms] and number of launches per kernel since the last reset */
int se_hip_get_timings(se_hip_pipeline* p, double ms_sum[SE_HIP_K_COUNT], int64_t launches[SE_HIP_K_COUNT], int32_t reset);
/* Kernel launches per kind since the last reset, counted on the host at enqueue time (no events, no synchronisation, does not flush a deferred
 * raycast): counts[SE_HIP_K_*], and counts[SE_HIP_K_COUNT] = launches that were the fused raycast + scan kernel (each of which also counts as one
 * raycast and one allocation scan).  Returns 1 if a deferred raycast is outstanding, else 0.  bench.py asserts with it that a timed region of K
 * frames holds K scans, K sweeps and K raycasts. */
int se_hip_get_launch_counts(se_hip_pipeline* p, int64_t counts[SE_HIP_K_COUNT + 1], int32_t reset);
/* Work counters behind the algorithmic-bytes figures of the roofline (instrumented kernel
 * variants, slower; off by default): out[0..7] = alloc probes, new keys, swept blocks, nodes,
 * get calls, interp calls, grad calls, ray hits -- accumulated since enabled / last read;
 * out[8..12] = raycast wave clocks (shader cycles): sum over waves of first-leaf search, march,
 * gradient+store, max wave lifetime, sum of LDS staging; out[13..15] reserved. */
int se_hip_enable_stats(se_hip_pipeline* p, int32_t on);
int se_hip_get_stats(se_hip_pipeline* p, uint64_t out[16], int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* SE_HIP_H */
