// facade_demo.cpp -- a caller written like the reference's build_map.cpp main loop
// (build_map.cpp:275-338) against the drop-in facade: Config::getInstance -> SurfelMapping ->
// processFrame per frame -> GlobalModel::downloadMap.  Frames come from a raw dump
// (u32 W,H,n; f32 fx,fy,cx,cy; then per frame rgb|depth|sem|pose16) written by the pytest.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../surfelmapping_amd/csrc/facade/SurfelMapping.h"

int main(int argc, char **argv)
{
    if (argc < 3) { std::printf("usage: facade_demo frames.bin out_map.bin\n"); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t hdr[3]; float intr[4];
    if (std::fread(hdr, 4, 3, f) != 3 || std::fread(intr, 4, 4, f) != 4) return 2;
    const int W = (int)hdr[0], H = (int)hdr[1], n = (int)hdr[2];
    Config::getInstance(intr[0], intr[1], intr[2], intr[3], H, W);          // build_map.cpp:282
    Config::maxSqrtVertices() = 1000;
    setenv("SM_PREPROCESS", "0", 0);
    SurfelMapping core;                                                      // build_map.cpp:286
    std::vector<unsigned char> rgb((size_t)W * H * 3), sem((size_t)W * H);
    std::vector<unsigned short> depth((size_t)W * H);
    for (int k = 0; k < n; ++k) {
        Eigen::Matrix4f pose;
        if (std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size() || std::fread(depth.data(), 2, depth.size(), f) != depth.size() ||
            std::fread(sem.data(), 1, sem.size(), f) != sem.size() || std::fread(pose.data(), 4, 16, f) != 16) return 2;
        core.processFrame(rgb.data(), depth.data(), sem.data(), &pose);    // build_map.cpp:301
        std::printf("frame %d: model %u offset %u data %u conflict %u unstable %u\n", k, core.getGlobalModel().getModel().second,
                    core.getGlobalModel().getOffset(), core.getGlobalModel().getData().second,
                    core.getGlobalModel().getConflict().second, core.getGlobalModel().getUnstable().second);
    }
    std::fclose(f);
    {   // what rungui() does with the core every frame (build_map.cpp:177-204): raw cloud, model, capacity pane
        pangolin::OpenGlMatrix mvp{}, mv{};
        core.getFeedbackBuffer(FeedbackBuffer::RAW)->render(mvp, core.getCurrPose(), false, true, false, false);
        core.getGlobalModel().renderModel(mvp, mv, 0.0, true, false, true, false, false, false, 3, 3);
        pangolin::GlTexture *nr = core.getGlobalModel().getModelMapNR();
        std::printf("raw cloud %u  drawn %zu  mirror %dx%d  textures %d %d\n", core.getFeedbackBuffer(FeedbackBuffer::RAW)->count(),
                    core.getGlobalModel().lastDrawnCount(), nr->width, nr->height, core.getTexture(GPUTexture::RGB)->width,
                    core.getTexture(GPUTexture::DEPTH_METRIC)->height);
    }
    if (argc > 4) {
        // what build_map.cpp:34-38 puts on the screen every frame: the RGB / DEPTH_METRIC / DEPTH_FILTERED textures (and LAST).
        // Without GL the handles point at host copies: dump them for the test
        FILE *t = std::fopen(argv[4], "wb");
        if (!t) return 2;
        const size_t P = (size_t)W * H;
        for (const char *n : {GPUTexture::DEPTH_METRIC, GPUTexture::DEPTH_FILTERED, "LAST"}) {
            pangolin::GlTexture *tx = core.getTexture(n);
            if (!tx->host || tx->width != W || tx->height != H) return 3;
            std::fwrite(tx->host, 4, P, t);
        }
        pangolin::GlTexture *tr = core.getTexture(GPUTexture::RGB), *ts = core.getTexture(GPUTexture::SEMANTIC), *td = core.getTexture(GPUTexture::DEPTH_RAW);
        if (!tr->host_u8 || !ts->host_u8 || !td->host_u16) return 3;
        std::fwrite(tr->host_u8, 1, P * 3, t); std::fwrite(ts->host_u8, 1, P, t); std::fwrite(td->host_u16, 2, P, t);
        std::fclose(t);
    }
    if (argc > 3) {                                                        // load_map.cpp-style novel-view dump
        std::vector<Eigen::Matrix4f> views = {core.getCurrPose()};
        core.acquireImages(argv[3], views, W, H, intr[0], intr[1], intr[2], intr[3], 7);
    }
    return core.getGlobalModel().downloadMap(argv[2], 0, n - 1) ? 0 : 1;  // build_map.cpp:254
}
