/*
 * HostVoxel.h — VVoxel / VVoxelVolume: the dense SDF grid container and its geometry rules.
 * Reference: Voxel/Public/Voxel.h:23-30, Voxel/Public/VoxelVolume.h:37-100,
 * Voxel/Private/VoxelVolume.cpp:19-27 (N = 2^r + 1, cell = 2*extent/(N-1)), :59-94 (Get/SetVoxel,
 * out-of-range ignored), :139-176 (index <-> position), :129-137 (dirty flag).
 */
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
#include "HostCore.h"

namespace VolumeRaytracer {
namespace Voxel {

struct VVoxel {
    uint8_t Material = 0;
    float Density = 30.f; /* Voxel.h:29 */
};
static_assert(sizeof(VVoxel) == 8, "VVoxel is 8 bytes on disk and on the wire (vrt_voxel)");

class VVoxelVolume {
public:
    VVoxelVolume(uint8_t resolution, float volumeExtends) { Reset(resolution, volumeExtends); }
    void Reset(uint8_t resolution, float volumeExtends) {
        Resolution = resolution;
        VolumeExtends = volumeExtends;
        VoxelCountAlongAxis = 2u + ((1u << resolution) - 1u);
        CellSize = (volumeExtends * 2) / ((float)VoxelCountAlongAxis - 1.f);
        Voxels.assign(GetVoxelCount(), VVoxel());
        DirtyFlag = true;
    }
    unsigned GetSize() const { return VoxelCountAlongAxis; }
    size_t GetVoxelCount() const { return (size_t)VoxelCountAlongAxis * VoxelCountAlongAxis * VoxelCountAlongAxis; }
    float GetVolumeExtends() const { return VolumeExtends; }
    float GetCellSize() const { return CellSize; }
    uint8_t GetResolution() const { return Resolution; }
    VAABB GetVolumeBounds() const { return VAABB(VVector::ZERO, VVector::ONE * VolumeExtends); }
    bool IsValidVoxelIndex(const VIntVector& i) const {
        const int n = (int)VoxelCountAlongAxis;
        return i.X >= 0 && i.X < n && i.Y >= 0 && i.Y < n && i.Z >= 0 && i.Z < n;
    }
    void SetVoxel(const VIntVector& i, const VVoxel& v) {
        if (IsValidVoxelIndex(i)) Voxels[VMathHelpers::Index3DTo1D(i.X, i.Y, i.Z, VoxelCountAlongAxis, VoxelCountAlongAxis)] = v;
    }
    VVoxel GetVoxel(const VIntVector& i) const {
        if (IsValidVoxelIndex(i)) return Voxels[VMathHelpers::Index3DTo1D(i.X, i.Y, i.Z, VoxelCountAlongAxis, VoxelCountAlongAxis)];
        return VVoxel();
    }
    void FillVolume(const VVoxel& v) {
        Voxels.assign(GetVoxelCount(), v);
        MakeDirty();
    }
    VVector VoxelIndexToRelativePosition(const VIntVector& i) const {
        return VVector((float)i.X, (float)i.Y, (float)i.Z) * CellSize + (-VVector::ONE * VolumeExtends);
    }
    VIntVector RelativePositionToCellIndex(const VVector& p) const {
        VVector r = p - (-VVector::ONE * VolumeExtends);
        return VIntVector((int)std::floor(r.X / CellSize), (int)std::floor(r.Y / CellSize), (int)std::floor(r.Z / CellSize));
    }
    VIntVector RelativePositionToVoxelIndex(const VVector& p) const {
        VVector r = p - (-VVector::ONE * VolumeExtends);
        return VIntVector((int)std::round(r.X / CellSize), (int)std::round(r.Y / CellSize), (int)std::round(r.Z / CellSize));
    }
    void SetMaterial(const VMaterial& m) { GeometryMaterial = m; }
    const VMaterial& GetMaterial() const { return GeometryMaterial; }
    void MakeDirty() { DirtyFlag = true; }
    bool IsDirty() const { return DirtyFlag; }
    void PostRender() { DirtyFlag = false; } /* VoxelVolume.cpp:114-117 */
    const std::vector<VVoxel>& GetVoxels() const { return Voxels; }
    std::vector<VVoxel>& GetVoxels() { return Voxels; }

    /* Metric of the density field for the sphere-trace (not part of the reference type): object-space
     * length of one density unit and the largest safe step; 1 / unbounded for analytic SDFs, the
     * extraction threshold / half of it for Voxelizer output. */
    float DensityScale = 1.f;
    float StepMax = 0.f;

private:
    std::vector<VVoxel> Voxels;
    float VolumeExtends = 0.f;
    float CellSize = 0.f;
    uint8_t Resolution = 0;
    unsigned VoxelCountAlongAxis = 0;
    VMaterial GeometryMaterial;
    bool DirtyFlag = true;
};

}  // namespace Voxel
}  // namespace VolumeRaytracer
