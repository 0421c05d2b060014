// SurfelMapping.h -- drop-in for src/SurfelMapping.h:18-128: same class, method names and
// argument order, backed by the HIP core through include/sm_c_api.h.
#pragma once
#include <cassert>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../../include/sm_c_api.h"
#include "Config.h"
#include "GPUTexture.h"
#include "GlobalModel.h"
#include "IndexMap.h"
#include "sm_compat.h"
#include "sm_png.h"
#include <sys/stat.h>

#include "FeedbackBuffer.h"

class Checker;          // debug aid of the reference (src/Utils/Checker.h); never constructed here

class SurfelMapping {
public:
    // src/SurfelMapping.cpp:12-25: reads the Config singleton, which must have been initialised
    // with real values first (build_map.cpp:282-286).
    SurfelMapping() : checker(nullptr)
    {
        sm_config c;
        sm_default_config(&c, Config::W(), Config::H(), Config::fx(), Config::fy(), Config::cx(), Config::cy());
        c.near_clip = Config::nearClip();
        c.far_clip = Config::farClip();
        c.fuse_thresh = Config::surfelFuseDistanceThreshFactor();
        c.max_sqrt_vertices = Config::maxSqrtVertices();
        const char *pre = std::getenv("SM_PREPROCESS");
        if (pre) c.preprocess = std::atoi(pre);
        // SM_FACADE_ASYNC=1: processFrame only enqueues (sm_process_frame_async: the images are staged inside the call, the copy of
        // frame f+1 overlaps frame f) and every getter waits -- the reference's processFrame ends in glFinish, so this is opt-in: a
        // device-side error is then reported by the next call that synchronises instead of by processFrame itself
        const char *as = std::getenv("SM_FACADE_ASYNC");
        async_ = as && as[0] == '1';
        ctx_ = sm_create(&c);
        if (!ctx_) throw std::runtime_error(std::string("SurfelMapping: ") + sm_last_error());
        globalModel.bind(ctx_);
        indexMap.bind(ctx_);
        rawFeedback.bind(ctx_);
        currPose = Eigen::Matrix4f::Identity();
        for (const char *n : {GPUTexture::RGB, GPUTexture::DEPTH_RAW, GPUTexture::DEPTH_FILTERED, GPUTexture::DEPTH_METRIC,
                              GPUTexture::SEMANTIC, "LAST"}) {
            textures[n] = new GPUTexture();
            textures[n]->texture->width = Config::W();
            textures[n]->texture->height = Config::H();
        }
    }
    virtual ~SurfelMapping()
    {
        for (auto &kv : textures) delete kv.second;
        sm_destroy(ctx_);
    }
    SurfelMapping(const SurfelMapping &) = delete;
    SurfelMapping &operator=(const SurfelMapping &) = delete;

    // src/SurfelMapping.h:31-34.  A null gtPose is an error here (the reference dereferences it:
    // src/SurfelMapping.cpp:130).
    void processFrame(const unsigned char *rgb, const unsigned short *depth = nullptr, const unsigned char *semantic = nullptr,
                      const Eigen::Matrix4f *gtPose = 0)
    {
        if (!gtPose) { std::printf("processFrame: gtPose is required\n"); return; }
        currPose = *gtPose;
        // the reference uploads the three images into its RGB / DEPTH / SEMANTIC textures (src/SurfelMapping.cpp:122-128; a null
        // depth / semantic keeps the old one): kept here as host copies for getTexture()
        const size_t P = (size_t)Config::W() * Config::H();
        textures[GPUTexture::RGB]->host_u8.assign(rgb, rgb + P * 3);
        if (depth) textures[GPUTexture::DEPTH_RAW]->host_u16.assign(depth, depth + P);
        if (semantic) textures[GPUTexture::SEMANTIC]->host_u8.assign(semantic, semantic + P);
        int rc = async_ ? sm_process_frame_async(ctx_, rgb, depth, semantic, gtPose->data())
                        : sm_process_frame(ctx_, rgb, depth, semantic, gtPose->data());
        if (rc != SM_OK) std::printf("processFrame: %s\n", sm_last_error());
        historyPoses.push_back(currPose);
    }
    // src/SurfelMapping.cpp:496-532
    void cleanPoints(const unsigned short *depth, const unsigned char *semantic, const Eigen::Matrix4f *gtPose)
    {
        currPose = *gtPose;
        if (sm_clean_points(ctx_, depth, semantic, gtPose->data()) != SM_OK) std::printf("cleanPoints: %s\n", sm_last_error());
        beginCleanPoints = false;
    }
    void reset() { sm_reset(ctx_); historyPoses.clear(); }          // src/SurfelMapping.cpp:436-441
    void setBeginCleanPoints() { beginCleanPoints = true; }
    bool getBeginCleanPoints() { return beginCleanPoints; }
    const Eigen::Matrix4f &getCurrPose() { return currPose; }
    const std::vector<Eigen::Matrix4f> &getHistoryPoses() { return historyPoses; }
    IndexMap &getIndexMap() { return indexMap; }
    GlobalModel &getGlobalModel() { return globalModel; }

    // src/SurfelMapping.cpp:450-456
    // The images live in the HIP context; the handle is filled when somebody asks for it (build_map.cpp:34-38 shows RGB,
    // DEPTH_METRIC and DEPTH_FILTERED every frame): the three float images are read back from the core (sm_download_depth),
    // the input images are the copies processFrame kept.  With SM_FACADE_GL the data goes into a real pangolin::GlTexture of the
    // reference's format (src/SurfelMapping.cpp:50-85); without GL the POD handle points at the host copy.
    pangolin::GlTexture *getTexture(const std::string &textureType)
    {
        auto it = textures.find(textureType);
        assert(it != textures.end() && "there is no such texture type");
        GPUTexture *t = it->second;
        const int W = Config::W(), H = Config::H();
        const int which = textureType == GPUTexture::DEPTH_METRIC ? SM_TEX_DEPTH_METRIC
                        : textureType == GPUTexture::DEPTH_FILTERED ? SM_TEX_DEPTH_FILTERED : textureType == "LAST" ? SM_TEX_LAST : -1;
        if (which >= 0) {
            t->host_f.resize((size_t)W * H);
            if (sm_download_depth(ctx_, which, t->host_f.data()) != SM_OK) std::printf("getTexture: %s\n", sm_last_error());
        }
#ifdef SM_FACADE_GL
        if (which >= 0) {
            if (!t->texture->tid) t->texture->Reinitialise(W, H, GL_R32F, false, 0, GL_RED, GL_FLOAT);
            t->texture->Upload(t->host_f.data(), GL_RED, GL_FLOAT);
        } else if (textureType == GPUTexture::RGB && !t->host_u8.empty()) {
            if (!t->texture->tid) t->texture->Reinitialise(W, H, GL_RGB32F, true, 0, GL_RGB, GL_UNSIGNED_BYTE);
            t->texture->Upload(t->host_u8.data(), GL_RGB, GL_UNSIGNED_BYTE);
        } else if (textureType == GPUTexture::DEPTH_RAW && !t->host_u16.empty()) {
            if (!t->texture->tid) t->texture->Reinitialise(W, H, GL_R16UI, false, 0, GL_RED_INTEGER, GL_UNSIGNED_SHORT);
            t->texture->Upload(t->host_u16.data(), GL_RED_INTEGER, GL_UNSIGNED_SHORT);
        } else if (textureType == GPUTexture::SEMANTIC && !t->host_u8.empty()) {
            if (!t->texture->tid) t->texture->Reinitialise(W, H, GL_R8UI, false, 0, GL_RED_INTEGER, GL_UNSIGNED_BYTE);
            t->texture->Upload(t->host_u8.data(), GL_RED_INTEGER, GL_UNSIGNED_BYTE);
        }
#elif defined(SM_COMPAT_POD_TEXTURE)
        t->texture->host = t->host_f.empty() ? nullptr : t->host_f.data();
        t->texture->host_u8 = t->host_u8.empty() ? nullptr : t->host_u8.data();
        t->texture->host_u16 = t->host_u16.empty() ? nullptr : t->host_u16.data();
#endif
        return t->texture;
    }
    // src/SurfelMapping.cpp:458-464: only "RAW" is ever created (src/SurfelMapping.cpp:88-90)
    FeedbackBuffer *getFeedbackBuffer(const std::string &feedbackType)
    {
        assert(feedbackType == FeedbackBuffer::RAW && "there is no such feedback buffer");
        (void)feedbackType;
        return &rawFeedback;
    }
    void computeFeedbackBuffers() { rawFeedback.refresh(); }                  // src/SurfelMapping.cpp:367-376

    // novel-view dump for SPADE (src/SurfelMapping.cpp:378-434): <path>/image/%06d.png (the bytes of the
    // reference's BGR cv::Mat, i.e. a correct-colour picture) and <path>/semantic/%06d.png (class + 1)
    void acquireImages(std::string path, const std::vector<Eigen::Matrix4f> &views, int w, int h, float fx, float fy, float cx,
                       float cy, int startId = 0)
    {
        if (path.empty() || path.back() != '/') path += "/";
        const std::string image_path = path + "image/", semantic_path = path + "semantic/";
        ::mkdir(image_path.c_str(), 0755);
        ::mkdir(semantic_path.c_str(), 0755);
        globalModel.setImageSize(w, h, fx, fy, cx, cy);
        std::vector<unsigned char> rgb((size_t)w * h * 3);
        for (const auto &v : views) {
            char name[32];
            std::snprintf(name, sizeof name, "%06d.png", startId);
            globalModel.renderImage(v);
            const std::vector<unsigned char> &bgr = globalModel.imageBGR();
            for (size_t p = 0; p < (size_t)w * h; ++p) { rgb[p * 3] = bgr[p * 3 + 2]; rgb[p * 3 + 1] = bgr[p * 3 + 1]; rgb[p * 3 + 2] = bgr[p * 3]; }
            const bool r1 = sm_png::write((image_path + name).c_str(), rgb.data(), w, h, 3);
            const bool r2 = sm_png::write((semantic_path + name).c_str(), globalModel.imageSemantic().data(), w, h, 1);
            if (!(r1 && r2)) std::printf("%s is NOT saved!\n", name);
            startId++;
        }
    }

    // extras of the HIP core
    sm_ctx *context() { return ctx_; }
    // (sm_sync first: with SM_FACADE_ASYNC frames may still be in flight, and sm_get_counts returns the counters of the last wait)
    sm_counts counts() { sm_counts c{}; (void)sm_sync(ctx_); sm_get_counts(ctx_, &c); return c; }

    Checker *checker;

private:
    sm_ctx *ctx_ = nullptr;
    Eigen::Matrix4f currPose;
    IndexMap indexMap;
    GlobalModel globalModel;
    FeedbackBuffer rawFeedback;
    std::map<std::string, GPUTexture *> textures;
    std::vector<Eigen::Matrix4f> historyPoses;
    bool beginCleanPoints = false;
    bool async_ = false;
};
