#include "HostCore.h"
namespace VolumeRaytracer {
const VVector VVector::ZERO(0.f, 0.f, 0.f);
const VVector VVector::ONE(1.f, 1.f, 1.f);
const VVector VVector::UP(0.f, 0.f, 1.f);
const VVector VVector::RIGHT(0.f, 1.f, 0.f);
const VVector VVector::FORWARD(1.f, 0.f, 0.f);
const VColor VColor::BLACK(0.f, 0.f, 0.f, 1.f);
const VColor VColor::WHITE(1.f, 1.f, 1.f, 1.f);
const VColor VColor::RED(1.f, 0.f, 0.f, 1.f);
const VColor VColor::GREEN(0.f, 1.f, 0.f, 1.f);
const VColor VColor::BLUE(0.f, 0.f, 1.f, 1.f);
const VQuat VQuat::IDENTITY(0.f, 0.f, 0.f, 1.f);
}  // namespace VolumeRaytracer
