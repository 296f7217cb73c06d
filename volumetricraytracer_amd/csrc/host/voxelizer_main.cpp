/* Voxelizer executable: `voxelizer [--gpu] [--out file.vox] path.gltf [texlib.json]` writes `<stem>.vox`
 * (Voxelizer/Private/Voxelizer.cpp:36-117).  --gpu runs the per-triangle loop on the first HIP device
 * (vrt_voxelize_mesh); the file is the same, byte for byte. */
#include <chrono>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/vrt.h"
#include "SceneConverter.h"
#include "VolumeConverter.h"

int main(int argc, char** argv) {
    bool gpu = false;
    std::string out;
    std::vector<std::string> args;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--gpu")) gpu = true;
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else args.push_back(argv[i]);
    }
    if (args.empty()) {
        std::cerr << "No file path for input file specified!" << std::endl;
        return 1;
    }
    vrt_ctx* ctx = nullptr;
    if (gpu) {
        const int rc = vrt_create(&ctx, 1, nullptr);
        if (rc != VRT_OK) {
            std::cerr << "[ERROR] --gpu: " << vrt_strerror(rc) << std::endl;
            return 1;
        }
        VolumeRaytracer::Voxelizer::VVolumeConverter::UseDevice(ctx);
    }
    int status = 0;
    try {
        const auto t0 = std::chrono::steady_clock::now();
        const std::string path = VolumeRaytracer::Voxelizer::VoxelizeFile(args[0], args.size() > 1 ? args[1] : "", out);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "Exported voxelized scene to: " << path << " (" << (gpu ? "device" : "host") << " voxelizer, " << s << " s)" << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "[ERROR] " << e.what() << std::endl;
        status = 1;
    }
    if (ctx) {
        VolumeRaytracer::Voxelizer::VVolumeConverter::UseDevice(nullptr);
        vrt_destroy(ctx);
    }
    return status;
}
