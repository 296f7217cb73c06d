/* vrt_launch.h — host-callable launch wrappers implemented in vrt_kernels.hip. */
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/vrt.h"
#include "vrt_device.h"

namespace vrt {

/* path: VRT_PATH_DENSE / VRT_PATH_BRICK / VRT_PATH_BRICK_LDS / kPathCube / kPathBrick16 / kPathCube16 (already resolved,
   never AUTO).
   block.f.n_frames frames (grid.y) in ONE launch, frame f with camera block.cam[f]. */
hipError_t launch_march(const DBlock& block, int path, bool single_instance, hipStream_t stream);
/* dense grid -> brick records of `format` (fp32: 512 B, VRT_FORMAT_TEXEL16: 256 B of int16). */
hipError_t launch_retile(const float* dense, void* bricks, int format, int N, int nb, hipStream_t stream);
/* VRT_PATH_CELLS: integer field -> nb^3 x 64 cell records of 8 int16. */
hipError_t launch_retile_cells16(const float* dense, void* cells, int N, int nb, hipStream_t stream);
/* VRT_FORMAT_TEXEL16: densities -> the integer field +-q of the reference's volume texel, in place. */
hipError_t launch_quantize_field(float* density, size_t count, hipStream_t stream);
/* The reference's RGBA8 volume texture (device copy) -> integer field + materials in the grid's own order. */
hipError_t launch_texels_to_field(const void* texels, float* density, uint8_t* material, int N, hipStream_t stream);
/* Empty-space table, level 1: nb^3 leap-count bytes from the dense grid (scratch: another nb^3 bytes), and (box6, six
   device ints) the bounding box of the near bricks in brick coordinates {min x, z, y, max x, z, y} ({nb.., -1..}: none). */
hipError_t launch_skip_table(const float* dense, uint8_t* table, uint8_t* scratch, int* box6, int N, int nb, float density_scale,
                             float step_max, hipStream_t stream);
/* Empty-space table, level 2: nb^3 words of sub-block nibbles (scratch: nibble_scratch_bytes(N)). */
size_t nibble_scratch_bytes(int N);
hipError_t launch_nibble_table(const float* dense, unsigned* nib, void* scratch, int N, int nb, float density_scale, float step_max,
                               hipStream_t stream);
/* Cube modes: nb^3-byte Chebyshev distance (bricks) to the nearest brick holding a solid voxel. */
hipError_t launch_cube_table(const float* dense, uint8_t* table, uint8_t* scratch, int N, int nb, hipStream_t stream);
/* Device Voxelizer: frames = n_frames vrt_vox::TriangleFrame records (device memory); writes N^3 densities + materials. */
hipError_t launch_voxelize(const void* frames, size_t n_frames, float* density, uint8_t* material, int N, float cell, float extent,
                           float threshold, hipStream_t stream);
/* vrt_debug_gather_ceiling: `blocks` workgroups of 256 lanes, `iters` trilinear samples per lane from a pool of n_bricks (a power of two)
   brick records of `format`; out: blocks * 256 floats. */
hipError_t launch_gather_ceiling(const void* pool, unsigned n_bricks, int format, bool coherent, int iters, float* out, int blocks, hipStream_t stream);
hipError_t launch_split_voxels(const void* voxels, float* density, uint8_t* material, size_t count,
                               hipStream_t stream);

}  // namespace vrt
