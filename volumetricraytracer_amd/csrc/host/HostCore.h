/*
 * HostCore.h — dependency-free restatement of the reference's Core math/value types that the
 * scene, the voxelizer and the .vox format need (the reference builds them on Eigen + boost).
 * Same names, members and semantics as the reference so that code written against its headers
 * reads the same; all arithmetic is plain fp32.
 *
 * Reference (relative to /root/reference/VolumetricRaytracer/VolumetricRaytracer/Core/):
 *   VVector / VIntVector / VVector2D   Public/Vector.h:20-154, Private/Vector.cpp
 *   VQuat                              Public/Quat.h:21-55, Private/Quat.cpp:28-121 (Eigen Quaternionf, x,y,z,w)
 *   VColor                             Public/Color.h, VAABB Public/AABB.h:22-44 + Private/AABB.cpp
 *   VMaterial                          Public/Material.h:22-42
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <string>

namespace VolumeRaytracer {

struct VVector {
    float X = 0.f, Y = 0.f, Z = 0.f;
    VVector() = default;
    VVector(float x, float y, float z) : X(x), Y(y), Z(z) {}
    VVector operator+(const VVector& o) const { return {X + o.X, Y + o.Y, Z + o.Z}; }
    VVector operator-(const VVector& o) const { return {X - o.X, Y - o.Y, Z - o.Z}; }
    VVector operator-() const { return {-X, -Y, -Z}; }
    VVector operator*(float s) const { return {X * s, Y * s, Z * s}; }
    VVector operator*(const VVector& o) const { return {X * o.X, Y * o.Y, Z * o.Z}; }
    VVector operator/(float s) const { return {X / s, Y / s, Z / s}; }
    float Dot(const VVector& o) const { return X * o.X + Y * o.Y + Z * o.Z; }
    VVector Cross(const VVector& o) const { return {Y * o.Z - Z * o.Y, Z * o.X - X * o.Z, X * o.Y - Y * o.X}; }
    static VVector Cross(const VVector& a, const VVector& b) { return a.Cross(b); }
    float LengthSquared() const { return X * X + Y * Y + Z * Z; }
    float Length() const { return std::sqrt(LengthSquared()); }
    VVector GetNormalized() const {
        float l = Length();
        return {X / l, Y / l, Z / l};
    }
    void Normalize() { *this = GetNormalized(); }
    VVector Abs() const { return {std::fabs(X), std::fabs(Y), std::fabs(Z)}; }
    static VVector Min(const VVector& a, const VVector& b) { return {std::fmin(a.X, b.X), std::fmin(a.Y, b.Y), std::fmin(a.Z, b.Z)}; }
    static VVector Max(const VVector& a, const VVector& b) { return {std::fmax(a.X, b.X), std::fmax(a.Y, b.Y), std::fmax(a.Z, b.Z)}; }
    static const VVector ZERO, ONE, UP, RIGHT, FORWARD; /* Vector.cpp:38-46: UP=+Z, RIGHT=+Y, FORWARD=+X */
};
static_assert(sizeof(VVector) == 12, "VVector is serialised as 12 bytes (VoxelObject.cpp:47)");

struct VIntVector {
    int X = 0, Y = 0, Z = 0;
    VIntVector() = default;
    VIntVector(int x, int y, int z) : X(x), Y(y), Z(z) {}
    VIntVector operator+(const VIntVector& o) const { return {X + o.X, Y + o.Y, Z + o.Z}; }
    VIntVector operator-(const VIntVector& o) const { return {X - o.X, Y - o.Y, Z - o.Z}; }
    bool operator==(const VIntVector& o) const { return X == o.X && Y == o.Y && Z == o.Z; }
};

struct VVector2D {
    float X = 0.f, Y = 0.f;
    VVector2D() = default;
    VVector2D(float x, float y) : X(x), Y(y) {}
};

struct VColor {
    float R = 0.f, G = 0.f, B = 0.f, A = 1.f;
    VColor() = default;
    VColor(float r, float g, float b, float a) : R(r), G(g), B(b), A(a) {}
    static const VColor BLACK, WHITE, RED, GREEN, BLUE;
};
static_assert(sizeof(VColor) == 16, "VColor is serialised as 16 bytes (Material.cpp:25)");

/* Unit quaternion stored x,y,z,w — the in-memory order of Eigen::Quaternionf, which is what the
 * reference memcpy's into .vox files (VoxelObject.cpp:49-55). */
struct VQuat {
    float x = 0.f, y = 0.f, z = 0.f, w = 1.f;
    VQuat() = default;
    VQuat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    static VQuat FromAxisAngle(const VVector& axis, float angle) { /* Quat.cpp:28-33 */
        VVector a = axis.GetNormalized();
        float s = std::sin(angle * 0.5f);
        return {a.X * s, a.Y * s, a.Z * s, std::cos(angle * 0.5f)};
    }
    static VQuat FromEulerAngles(float roll, float yaw, float pitch) { /* :57-60 */
        return FromAxisAngle(VVector::RIGHT, pitch) * FromAxisAngle(VVector::UP, yaw) * FromAxisAngle(VVector::FORWARD, roll);
    }
    VQuat operator*(const VQuat& b) const { /* Hamilton product: apply b first, then *this */
        return {w * b.x + x * b.w + y * b.z - z * b.y, w * b.y - x * b.z + y * b.w + z * b.x,
                w * b.z + x * b.y - y * b.x + z * b.w, w * b.w - x * b.x - y * b.y - z * b.z};
    }
    VVector operator*(const VVector& v) const { /* rotate, :91-95 */
        VVector q(x, y, z);
        VVector uv = q.Cross(v) * 2.0f;
        return v + uv * w + q.Cross(uv);
    }
    VQuat Inverse() const {
        float n = x * x + y * y + z * z + w * w;
        return {-x / n, -y / n, -z / n, w / n};
    }
    VVector GetForwardVector() const { return *this * VVector::FORWARD; }
    VVector GetRightVector() const { return *this * VVector::RIGHT; }
    VVector GetUpVector() const { return *this * VVector::UP; }
    float GetX() const { return x; }
    float GetY() const { return y; }
    float GetZ() const { return z; }
    float GetW() const { return w; }
    static const VQuat IDENTITY;
};
static_assert(sizeof(VQuat) == 16, "VQuat is serialised as 16 bytes");

struct VAABB {
    VVector Position;
    VVector Extends = VVector(0.5f, 0.5f, 0.5f);
    VAABB() = default;
    VAABB(const VVector& position, const VVector& extends) : Position(position), Extends(extends) {}
    void SetCenterPosition(const VVector& p) { Position = p; }
    void SetExtends(const VVector& e) { Extends = e.Abs(); }
    VVector GetMin() const { return Position - Extends; }
    VVector GetMax() const { return Position + Extends; }
    VVector GetExtends() const { return Extends; }
    VVector GetCenterPosition() const { return Position; }
};

struct VMaterial {
    VColor AlbedoColor = VColor(0.8f, 0.8f, 0.8f, 1.f);
    float Roughness = 0.8f;
    float Metallic = 0.f;
    std::string AlbedoTexturePath, NormalTexturePath, RMTexturePath;
    VVector2D TextureScale = VVector2D(100.f, 100.f);
};

namespace VMathHelpers {
inline float ToRadians(float degrees) { return degrees * (3.141592f / 180.f); } /* MathHelpers (2).cpp:38-41 */
inline size_t Index3DTo1D(int x, int y, int z, size_t yCount, size_t zCount) { return (size_t)x * yCount * zCount + (size_t)z * yCount + (size_t)y; } /* :43-46 */
inline void Index1DTo3D(size_t index, size_t yCount, size_t zCount, int& x, int& y, int& z) { /* :26-31 */
    x = (int)(index / (yCount * zCount));
    z = (int)((index - (size_t)x * yCount * zCount) / yCount);
    y = (int)(index - (size_t)x * yCount * zCount - (size_t)z * yCount);
}
template <typename T> inline T Clamp(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace VMathHelpers

}  // namespace VolumeRaytracer
