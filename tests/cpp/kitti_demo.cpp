// kitti_demo.cpp -- build_map.cpp's main loop (build_map.cpp:275-338) on a KITTI-layout directory through the drop-in
// KittiReader + SurfelMapping facade: reader -> Config -> core -> while(getNext()) processFrame -> downloadMap.
// With a single argument it only decodes the frames and prints checksums (no GPU needed).
#include <cstdio>
#include <cstdlib>
#include "../../surfelmapping_amd/csrc/facade/KittiReader.h"
#include "../../surfelmapping_amd/csrc/facade/SurfelMapping.h"

static unsigned long long fnv(const void *p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= ((const unsigned char *)p)[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::printf("usage: kitti_demo <dataset dir> [out_map.bin]\n"); return 2; }
    KittiReader reader(argv[1], false, true, 0, true);                       // build_map.cpp:279
    if (!reader.good()) { std::printf("bad dataset\n"); return 3; }
    std::printf("calib %g %g %g %g %d %d frames %zu\n", reader.fx(), reader.fy(), reader.cx(), reader.cy(), reader.W(), reader.H(),
                reader.numFrames());
    if (argc == 2) {
        reader.setState(-1);
        while (reader.getNext())
            std::printf("frame %d t=%.3f rgb %016llx depth %016llx sem %016llx pose %016llx\n", reader.currentFrameId, reader.time,
                        fnv(reader.rgb, (size_t)reader.numPixels() * 3), fnv(reader.depth, (size_t)reader.numPixels() * 2),
                        fnv(reader.semantic, (size_t)reader.numPixels()), fnv(reader.gtPose.data(), 64));
        return 0;
    }
    Config::getInstance(reader.fx(), reader.fy(), reader.cx(), reader.cy(), reader.H(), reader.W());   // :282
    Config::maxSqrtVertices() = 1000;
    setenv("SM_PREPROCESS", "1", 0);
    SurfelMapping core;                                                      // :286
    reader.setState(-1);                                                     // first getNext() yields frame 0
    while (reader.getNext()) {                                               // :294
        core.processFrame(reader.rgb, reader.depth, reader.semantic, &reader.gtPose);   // :301
        std::printf("frame %d: model %u\n", reader.currentFrameId, core.getGlobalModel().getModel().second);
    }
    return core.getGlobalModel().downloadMap(argv[2], 0, (int)reader.numFrames() - 1) ? 0 : 1;
}
