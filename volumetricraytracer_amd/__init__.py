"""MI355X-native volumetric SDF ray-marcher behind the VRenderer surface of
Elyptos/VolumetricRaytracer.  The product is the C-ABI library built from csrc/ (see
include/vrt.h); this package is its host-side mirror of the reference's scene types."""
from . import _abi
from .renderer import VHipRenderer, algorithmic_bytes
from .scene import (ADD, FORWARD, IDENTITY, RIGHT, SUBTRACT, UP, VBox, VCamera, VCylinder, VDensityGenerator,
                    VLight, VMaterial, VPointLight, VScene, VSphere, VSpotLight, VVoxelObject, VVoxelVolume,
                    csg_volume, default_params, demo_light, march_budget, look_minus_x_camera, procedural_skybox,
                    quat_from_axis_angle, quat_from_euler_deg, quat_inverse, quat_mul, quat_rotate, sphere_volume,
                    torus_volume)

__all__ = [n for n in dir() if not n.startswith("_")] + ["_abi"]
