/* Voxelizer executable: `voxelizer path.gltf [texlib.json]` writes `<stem>.vox`
 * (Voxelizer/Private/Voxelizer.cpp:36-117). */
#include <iostream>

#include "SceneConverter.h"

int main(int argc, char** argv) {
    if (argc <= 1) {
        std::cerr << "No file path for input file specified!" << std::endl;
        return 1;
    }
    try {
        const std::string out = VolumeRaytracer::Voxelizer::VoxelizeFile(argv[1], argc > 2 ? argv[2] : "", "");
        std::cout << "Exported voxelized scene to: " << out << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "[ERROR] " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
