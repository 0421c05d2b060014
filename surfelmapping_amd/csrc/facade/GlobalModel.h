// GlobalModel.h -- drop-in for src/GlobalModel.h:16-120 over the C-ABI.  The per-pass methods
// keep the reference's names and call order (src/SurfelMapping.cpp:178-239); passes that the
// HIP core fuses into a neighbour are no-ops here (noted per method).
#pragma once
#include <cstdio>
#include <string>
#include <utility>
#include <vector>
#include "../../../include/sm_c_api.h"
#include "Config.h"
#include "GPUTexture.h"
#include "sm_compat.h"

class GlobalModel {
public:
    explicit GlobalModel(sm_ctx *ctx = nullptr)
        : TEXTURE_DIMENSION(Config::maxSqrtVertices()), MAX_VERTICES(TEXTURE_DIMENSION * TEXTURE_DIMENSION), ctx_(ctx) {}
    void bind(sm_ctx *ctx) { ctx_ = ctx; }

    const int TEXTURE_DIMENSION;
    const int MAX_VERTICES;

    // p2 (+p3): src/GlobalModel.cpp:396-515
    void processConflict(const Eigen::Matrix4f &pose, const int & /*time*/, GPUTexture * /*depthRaw*/, GPUTexture * /*semantic*/,
                         float minDepth, float maxDepth, float fuseThresh = Config::surfelFuseDistanceThreshFactor(), int isClean = 0)
    {
        if (sm_stage_conflict(ctx_, pose.data(), minDepth, maxDepth, fuseThresh, isClean) == SM_OK) pending_ = true;
        else std::printf("processConflict: %s\n", sm_last_error());
    }
    void updateConflict() {}                       // in-place decrement, applied by backMapping()
    // p4/p10: src/GlobalModel.cpp:517-579 (the 2nd call per frame copies nothing: fuse is in place)
    void backMapping()
    {
        if (pending_) { if (sm_stage_cull(ctx_) != SM_OK) std::printf("backMapping: %s\n", sm_last_error()); pending_ = false; }
    }
    void buildModelMap() {}                        // no mirror textures (src/GlobalModel.cpp:639-681)
    // p8 + p9 + p11: src/GlobalModel.cpp:246-394,581-637
    void dataAssociate(const Eigen::Matrix4f &pose, const int &time, GPUTexture *, GPUTexture *, GPUTexture *, GPUTexture *,
                       GPUTexture *, GPUTexture *, GPUTexture *, float depthMin, float depthMax)
    {
        int rc = sm_stage_associate_fuse(ctx_, pose.data(), time, depthMin, depthMax);
        if (rc != SM_OK) std::printf("dataAssociate: %s\n", sm_last_error());
    }
    void updateFuse() {}
    void concatenate() {}

    std::pair<GLuint, GLuint> getModel() { return {0u, counts().count}; }
    std::pair<GLuint, GLuint> getData() { return {0u, counts().data_count}; }
    std::pair<GLuint, GLuint> getConflict() { return {0u, counts().conflict_count}; }
    std::pair<GLuint, GLuint> getUnstable() { return {0u, counts().unstable_count}; }
    unsigned int getOffset() { return counts().offset; }

    // src/GlobalModel.cpp:901-1011, same file format and diagnostics
    bool downloadMap(const std::string &path, int startId, int endId)
    {
        if (sm_save_map(ctx_, path.c_str(), startId, endId) != SM_OK) { std::printf("%s\n", sm_last_error()); return false; }
        std::printf("%s is saved! Saved model count: %d\n", path.c_str(), (int)counts().count);
        return true;
    }
    bool uploadMap(const std::string &model_path, std::vector<int> &start_end_ids)
    {
        int32_t a = 0, b = 0;
        if (sm_load_map(ctx_, model_path.c_str(), &a, &b) != SM_OK) { std::printf("%s\n", sm_last_error()); return false; }
        std::printf("Load model count: %d\nRead model from %s.\n", (int)counts().count, model_path.c_str());
        start_end_ids.clear(); start_end_ids.push_back(a); start_end_ids.push_back(b);
        return true;
    }
    void resetBuffer() { sm_reset(ctx_); }

    // model read-back in the reference's AoS layout (12 floats / surfel, src/Config.cpp:17-32)
    std::vector<float> downloadModel()
    {
        uint32_t n = 0;
        sm_download_model_aos(ctx_, nullptr, 0, &n);
        std::vector<float> v((size_t)n * 12);
        if (n) sm_download_model_aos(ctx_, v.data(), n, &n);
        return v;
    }

    // novel views for SPADE (src/GlobalModel.cpp:772-833): rendered by the HIP core, kept on the host
    void setImageSize(int w, int h, float fx, float fy, float cx, float cy)
    {
        iw_ = w; ih_ = h; ifx_ = fx; ify_ = fy; icx_ = cx; icy_ = cy;
        imageBgr_.assign((size_t)w * h * 3, 0);
        imageSem_.assign((size_t)w * h, 0);
        imageTex_.texture->width = semTex_.texture->width = w;
        imageTex_.texture->height = semTex_.texture->height = h;
    }
    void renderImage(const Eigen::Matrix4f &view)
    {
        if (sm_render_image(ctx_, view.data(), iw_, ih_, ifx_, ify_, icx_, icy_, imageBgr_.data(), imageSem_.data()) != SM_OK)
            std::printf("renderImage: %s\n", sm_last_error());
    }
    pangolin::GlTexture *getImageTex() { return imageTex_.texture; }
    pangolin::GlTexture *getSemanticTex() { return semTex_.texture; }
    const std::vector<unsigned char> &imageBGR() const { return imageBgr_; }      // h*w*3, B,G,R (FragColor = srgb.wzy)
    const std::vector<unsigned char> &imageSemantic() const { return imageSem_; } // h*w, class + 1, 0 = empty
    int imageWidth() const { return iw_; }
    int imageHeight() const { return ih_; }

    // src/GlobalModel.cpp:683-758 (signature src/GlobalModel.h:27-37).  The compute core keeps no GL buffer: the model is
    // pulled to the host when somebody looks (the reference's AoS layout, 12 floats per surfel) and, built with
    // SM_FACADE_GL, drawn as GL_POINTS; `threshold` / `time` / `timeDelta` select what the reference's shader discards
    // (confidence below threshold unless drawUnstable, last seen more than timeDelta frames ago) -- applied on the host copy.
    void renderModel(pangolin::OpenGlMatrix mvp, pangolin::OpenGlMatrix mv, float threshold, bool drawUnstable, bool drawNormals,
                     bool drawColors, bool drawPoints, bool drawWindow, bool drawSemantic, int time, int timeDelta)
    {
        (void)mv; (void)drawNormals; (void)drawColors; (void)drawPoints; (void)drawWindow; (void)drawSemantic; (void)time; (void)timeDelta;
        refreshHostModel();
        drawn_.clear();
        const size_t n = hostModel_.size() / 12;
        for (size_t k = 0; k < n; ++k) {
            const float *v = &hostModel_[k * 12];
            if (!drawUnstable && v[3] < threshold) continue;                 // draw_surface.vert: confidence gate
            drawn_.insert(drawn_.end(), v, v + 3);
        }
#ifdef SM_FACADE_GL
        glMatrixMode(GL_PROJECTION); glLoadIdentity(); glMultMatrixd(mvp.m);
        glMatrixMode(GL_MODELVIEW); glLoadIdentity();
        glEnableClientState(GL_VERTEX_ARRAY);
        glVertexPointer(3, GL_FLOAT, 0, drawn_.data());
        glDrawArrays(GL_POINTS, 0, (GLsizei)(drawn_.size() / 3));
        glDisableClientState(GL_VERTEX_ARRAY);
#else
        (void)mvp;
#endif
    }
    size_t lastDrawnCount() const { return drawn_.size() / 3; }

    // The model-aligned mirror textures (src/GlobalModel.cpp:639-681) do not exist in the compute core.  GUI::drawCapacity
    // (build_map.cpp:204) shows the fill level of the TEXTURE_DIMENSION^2 normal/radius mirror: the handle is filled lazily --
    // size TEXTURE_DIMENSION x TEXTURE_DIMENSION, and (without GL) the host copy of the plane for whoever wants to look.
    pangolin::GlTexture *getModelMapVC() { return fillMirror(mapVC_, 0); }
    pangolin::GlTexture *getModelMapCT() { return fillMirror(mapCT_, 4); }
    pangolin::GlTexture *getModelMapNR() { return fillMirror(mapNR_, 8); }
    const std::vector<float> &mirrorHost(int which) const { return mirror_[which]; }      // 0 VC, 1 CT, 2 NR: count x 4 floats

private:
    void refreshHostModel() { hostModel_ = downloadModel(); }
    pangolin::GlTexture *fillMirror(GPUTexture &t, int off)
    {
        refreshHostModel();
        std::vector<float> &m = mirror_[off / 4];
        const size_t n = hostModel_.size() / 12;
        m.resize(n * 4);
        for (size_t k = 0; k < n; ++k)
            for (int c = 0; c < 4; ++c) m[k * 4 + c] = hostModel_[k * 12 + off + c];
        if (off == 4) for (size_t k = 0; k < n; ++k) m[k * 4 + 1] = (float)k;            // map.vert:32 stores float(index) in .y
        t.texture->width = TEXTURE_DIMENSION;
        t.texture->height = TEXTURE_DIMENSION;
#ifdef SM_FACADE_GL
        if (!t.texture->tid) t.texture->Reinitialise(TEXTURE_DIMENSION, TEXTURE_DIMENSION, GL_RGBA32F, false, 0, GL_RGBA, GL_FLOAT);
        const int rows = (int)((n + TEXTURE_DIMENSION - 1) / TEXTURE_DIMENSION);
        if (rows) { m.resize((size_t)rows * TEXTURE_DIMENSION * 4, 0.0f); t.texture->Upload(m.data(), 0, 0, TEXTURE_DIMENSION, rows, GL_RGBA, GL_FLOAT); }
#endif
        return t.texture;
    }
    // (sm_sync first: with SM_FACADE_ASYNC frames may still be in flight, and sm_get_counts returns the counters of the last wait)
    sm_counts counts() { sm_counts c{}; (void)sm_sync(ctx_); sm_get_counts(ctx_, &c); return c; }
    std::vector<float> hostModel_, drawn_, mirror_[3];
    sm_ctx *ctx_;
    bool pending_ = false;
    GPUTexture mapVC_, mapCT_, mapNR_, imageTex_, semTex_;
    int iw_ = 0, ih_ = 0;
    float ifx_ = 0, ify_ = 0, icx_ = 0, icy_ = 0;
    std::vector<unsigned char> imageBgr_, imageSem_;
};
