/* vrt_launch.h — host-callable launch wrappers implemented in vrt_kernels.hip. */
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/vrt.h"
#include "vrt_device.h"

namespace vrt {

/* path: VRT_PATH_DENSE / VRT_PATH_BRICK / VRT_PATH_BRICK_LDS / kPathCube (already resolved, never AUTO). */
hipError_t launch_march(const DFrame& frame, int path, bool single_instance, hipStream_t stream);
hipError_t launch_retile(const float* dense, float* bricks, int N, int nb, hipStream_t stream);
/* Builds the nb^3-byte empty-space table from the brick pool (scratch: another nb^3 bytes). */
hipError_t launch_skip_table(const float* bricks, uint8_t* table, uint8_t* scratch, int nb, float density_scale, float step_max,
                             hipStream_t stream);
/* Cube modes: nb^3-byte Chebyshev distance (bricks) to the nearest brick holding a solid voxel. */
hipError_t launch_cube_table(const float* bricks, uint8_t* table, uint8_t* scratch, int N, int nb, hipStream_t stream);
/* Device Voxelizer: frames = n_frames vrt_vox::TriangleFrame records (device memory); writes N^3 densities + materials. */
hipError_t launch_voxelize(const void* frames, size_t n_frames, float* density, uint8_t* material, int N, float cell, float extent,
                           float threshold, hipStream_t stream);
hipError_t launch_split_voxels(const void* voxels, float* density, uint8_t* material, size_t count,
                               hipStream_t stream);

}  // namespace vrt
