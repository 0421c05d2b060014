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

class Checker;          // debug aid of the reference (src/Utils/Checker.h); never constructed here
class FeedbackBuffer;   // raw per-frame cloud (GUI "Draw raw"); not produced by the compute core

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
        ctx_ = sm_create(&c);
        if (!ctx_) throw std::runtime_error(std::string("SurfelMapping: ") + sm_last_error());
        globalModel.bind(ctx_);
        indexMap.bind(ctx_);
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
        int rc = sm_process_frame(ctx_, rgb, depth, semantic, gtPose->data());
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
    pangolin::GlTexture *getTexture(const std::string &textureType)
    {
        auto it = textures.find(textureType);
        assert(it != textures.end() && "there is no such texture type");
        return it->second->texture;
    }
    FeedbackBuffer *getFeedbackBuffer(const std::string &) { return nullptr; }

    // novel-view dump for SPADE (src/SurfelMapping.cpp:378-434): SURVEY 8f rank 3, not built
    void acquireImages(std::string, const std::vector<Eigen::Matrix4f> &, int, int, float, float, float, float, int = 0)
    {
        std::printf("acquireImages: novel-view renderer is not part of the compute core yet\n");
    }

    // extras of the HIP core
    sm_ctx *context() { return ctx_; }
    sm_counts counts() { sm_counts c{}; sm_get_counts(ctx_, &c); return c; }

    Checker *checker;

private:
    sm_ctx *ctx_ = nullptr;
    Eigen::Matrix4f currPose;
    IndexMap indexMap;
    GlobalModel globalModel;
    std::map<std::string, GPUTexture *> textures;
    std::vector<Eigen::Matrix4f> historyPoses;
    bool beginCleanPoints = false;
};
