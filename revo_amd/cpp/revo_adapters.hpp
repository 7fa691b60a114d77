// revo_adapters.hpp -- header-only C++ adapters over the C ABI (include/revo_hip.h).
//
// Re-creates the reference's class names and method surface so a system.cpp-style
// host links against librevo_hip.so instead of datastructures/imgpyramidrgbd.cpp,
// system/tracker.cpp, system/optimizer.cpp and utils/LGSX.h:
//
//   ImgPyramidSettings / Camera / CameraPyr   datastructures/camerapyr.h:27-193
//   ImgPyramidRGBD                            datastructures/imgpyramidrgbd.h:27-250
//   OptimizerSettings / Optimizer             system/optimizer.h:42-186
//   TrackerSettings / TrackerNew              system/tracker.h:31-105
//
// Reference-typed accessors.  A host that has Eigen and OpenCV (the reference does) defines, before including this
// header,
//     #define REVO_MATXF Eigen::MatrixXf      // needs resize(rows, cols), data(), cols()
//     #define REVO_VEC4F Eigen::Vector4f      // 16 bytes
//     #define REVO_CVMAT cv::Mat              // needs create(rows, cols, type), data, step; CV_8UC1 / CV_32FC1 defined
// and gets the reference's own signatures (imgpyramidrgbd.h:45-117, tracker.h:69-80):
//     const Eigen::MatrixXf& return3DEdges(lvl), const Eigen::Vector4f* returnOptimizationStructure(lvl),
//     const cv::Mat& returnDistTransform / returnEdges / returnOrigEdges / returnDepth / returnGray(lvl),
//     TrackerNew(const TrackerSettings&, const ImgPyramidSettings&), addOldPclAndPose(const Eigen::MatrixXf&, pose, ts)
// -- cached host mirrors of the device planes, returned by const reference like the reference's members.
// tests/cpp/adapter_refshape.cpp compiles them against Eigen / cv shaped stand-ins and static_asserts the types.
//
// No Eigen / OpenCV dependency otherwise: matrices are passed as anything with .data()
// (Eigen::Matrix3f / Vector3f / Matrix4f are column-major, which is what the ABI
// wants) and images as anything with .data and .step (cv::Mat) or raw pointers.
// Error behaviour follows the reference: a failing call logs and exit(0)s
// (imgpyramidrgbd.h:115-116, camerapyr.h:34-38), a non-orthogonal R abort()s
// (Sophus ENSURE, common.hpp:117-136).  Define REVO_ADAPTERS_THROW to get
// std::runtime_error instead.
#pragma once
#include <algorithm>
#include <array>
#include <map>
#include <exception>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/revo_hip.h"

namespace revo {

inline void fatal(int rc, const char* what) {
#ifdef REVO_ADAPTERS_THROW
  throw std::runtime_error(std::string(what) + ": " + revo_last_error());
#else
  std::fprintf(stderr, "[revo] %s: %s\n", what, revo_last_error());
  if (rc == REVO_ERR_NOT_ORTHOGONAL) std::abort();  // Sophus SOPHUS_ENSURE
  std::exit(0);                                     // the reference's I3D_LOG(error) + exit(0)
#endif
}
inline void check(int rc, const char* what) { if (rc != REVO_OK) fatal(rc, what); }

// ---- settings ------------------------------------------------------------------
struct ImgPyramidSettings : revo_pyr_settings {  // camerapyr.h:27-89
  ImgPyramidSettings() { revo_pyr_settings_default(this); }
  int nLevels() const { return pyr_min_lvl - pyr_max_lvl + 1; }  // camerapyr.h:68-71
  int& PYR_MIN_LVL() { return pyr_min_lvl; }
  int& PYR_MAX_LVL() { return pyr_max_lvl; }
};
struct OptimizerSettings : revo_opt_settings {  // optimizer.h:42-112
  OptimizerSettings() { revo_opt_settings_default(this); }
};
struct TrackerSettings : revo_tracker_settings {  // tracker.h:31-55
  OptimizerSettings optimizerSettings;
  TrackerSettings() { revo_tracker_settings_default(this); }
};

struct Camera {  // camerapyr.h:90-111
  float fx, fy, cx, cy;
  size_t width, height, area;
};

// ---- CameraPyr: intrinsics per level + owner of the device context -------------
class CameraPyr {  // camerapyr.h:113-193
 public:
  explicit CameraPyr(const ImgPyramidSettings& settingsPyr, int device = 0) {
    check(revo_ctx_create(device, &settingsPyr, nullptr, nullptr, &ctx_), "CameraPyr");
    for (int lvl = 0; lvl < settingsPyr.nLevels(); ++lvl) {
      float k[6];
      check(revo_ctx_camera(ctx_, lvl, k), "CameraPyr::at");
      camPyr.push_back(Camera{k[0], k[1], k[2], k[3], (size_t)k[4], (size_t)k[5], (size_t)k[4] * (size_t)k[5]});
    }
  }
  ~CameraPyr() { revo_ctx_destroy(ctx_); }
  CameraPyr(const CameraPyr&) = delete;
  CameraPyr& operator=(const CameraPyr&) = delete;
  int size() const { return (int)camPyr.size(); }
  const Camera& at(int lvl) const { return camPyr.at(lvl); }
  revo_ctx* ctx() const { return ctx_; }
  std::vector<Camera> camPyr;

 private:
  revo_ctx* ctx_ = nullptr;
};

// host mirrors handed out by return3DEdges(): addOldPclAndPose(matrix, ...) recognises them and copies the cloud on
// the device instead of uploading it again
struct MirrorRegistry {
  std::mutex mu;
  std::map<const float*, std::pair<revo_pyr*, int>> by_data;
  static MirrorRegistry& get() { static MirrorRegistry r; return r; }
  void add(const float* p, revo_pyr* pyr, int lvl) { std::lock_guard<std::mutex> lk(mu); by_data[p] = {pyr, lvl}; }
  void drop(revo_pyr* pyr) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto it = by_data.begin(); it != by_data.end();) it = (it->second.first == pyr) ? by_data.erase(it) : std::next(it);
  }
  bool find(const float* p, revo_pyr** pyr, int* lvl) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = by_data.find(p);
    if (it == by_data.end()) return false;
    *pyr = it->second.first; *lvl = it->second.second;
    return true;
  }
};

// ---- ImgPyramidRGBD ---------------------------------------------------------------
class ImgPyramidRGBD {  // imgpyramidrgbd.h:27-250
 public:
  // raw-pointer form of ImgPyramidRGBD(settings, camPyr, fullResRgb, fullResDepth, timestamp)
  ImgPyramidRGBD(const ImgPyramidSettings& settings, const std::shared_ptr<CameraPyr>& camPyr, const uint8_t* bgr,
                 size_t bgr_stride, const float* depth_m, size_t depth_stride, double timestamp)
      : cameraPyr(camPyr), mSettings(settings) {
    check(revo_pyramid_create(camPyr->ctx(), bgr, bgr_stride, depth_m, depth_stride, timestamp, &pyr_), "ImgPyramidRGBD");
    setIdentity();
  }
  // cv::Mat-like form (anything with .data and .step): BGR8 + CV_32FC1 metres
  template <class MatRGB, class MatDepth>
  ImgPyramidRGBD(const ImgPyramidSettings& settings, const std::shared_ptr<CameraPyr>& camPyr, const MatRGB& fullResRgb,
                 const MatDepth& fullResDepth, double timestamp)
      : ImgPyramidRGBD(settings, camPyr, (const uint8_t*)fullResRgb.data, (size_t)fullResRgb.step,
                       (const float*)fullResDepth.data, (size_t)fullResDepth.step, timestamp) {}
  ~ImgPyramidRGBD() { MirrorRegistry::get().drop(pyr_); revo_pyramid_destroy(pyr_); }
  ImgPyramidRGBD(const ImgPyramidRGBD&) = delete;
  ImgPyramidRGBD& operator=(const ImgPyramidRGBD&) = delete;

  void makeKeyframe() { check(revo_pyramid_make_keyframe(pyr_), "makeKeyframe"); }  // imgpyramidrgbd.cpp:231-252
  void prepareKfForStorage() {}                                                      // imgpyramidrgbd.h:156-169 (no-op)

#if defined(REVO_MATXF) && defined(REVO_VEC4F) && defined(REVO_CVMAT)
  // ---- the reference's accessor signatures (imgpyramidrgbd.h:45-117): host mirrors, filled on first use
  const REVO_MATXF& return3DEdges(unsigned lvl) const {  // 4 x N, column-major
    auto& m = slot(m3d_, lvl);
    if (!m) {
      const std::vector<float> v = readF(REVO_PLANE_EDGES3D, lvl, 4);
      m.reset(new REVO_MATXF());
      m->resize(4, (long)(v.size() / 4));
      std::copy(v.begin(), v.end(), m->data());
      MirrorRegistry::get().add(m->data(), pyr_, (int)lvl);
    }
    return *m;
  }
  const REVO_VEC4F* returnOptimizationStructure(unsigned lvl) const {  // (-dDT/dx, -dDT/dy, DT, 0) per pixel
    static_assert(sizeof(REVO_VEC4F) == 16, "REVO_VEC4F must be four packed floats");
    auto& t = slot(tab_, lvl);
    if (!t) {
      const std::vector<float> v = readF(REVO_PLANE_GRADTABLE, lvl, 4);  // fails like "optimizationStructure not built!"
      t.reset(new std::vector<REVO_VEC4F>(v.size() / 4));
      std::memcpy((void*)t->data(), v.data(), v.size() * sizeof(float));
    }
    return t->data();
  }
  const REVO_CVMAT& returnDistTransform(unsigned lvl) const { return matF(dt_, REVO_PLANE_DT, lvl); }
  const REVO_CVMAT& returnDepth(unsigned lvl) const { return matF(depth_, REVO_PLANE_DEPTH, lvl); }
  const REVO_CVMAT& returnEdges(unsigned lvl) const { return matU8(edges_, REVO_PLANE_EDGES, lvl); }
  const REVO_CVMAT& returnOrigEdges(unsigned lvl) const { return matU8(orig_, REVO_PLANE_EDGES_ORIG, lvl); }
  const REVO_CVMAT& returnGray(unsigned lvl) const { return matU8(gray_, REVO_PLANE_GRAY, lvl); }
  std::vector<float> generateColoredPcl(unsigned lvl, bool densePcl) const {
    size_t n = 0;
    check(revo_pyramid_colored_pcl(pyr_, (int)lvl, densePcl ? 1 : 0, nullptr, 0, &n), "generateColoredPcl");
    std::vector<float> v(n * 8);
    if (n) check(revo_pyramid_colored_pcl(pyr_, (int)lvl, densePcl ? 1 : 0, v.data(), n, &n), "generateColoredPcl");
    return v;
  }
  template <class MatX>
  void generateColoredPcl(unsigned lvl, MatX& clrPcl, bool densePcl) const {  // the reference's signature (Eigen::MatrixXf)
    const std::vector<float> v = generateColoredPcl(lvl, densePcl);
    clrPcl.resize(8, (long)(v.size() / 8));
    std::copy(v.begin(), v.end(), clrPcl.data());
  }
#else
  // accessors: lazy device->host copies (imgpyramidrgbd.h:45-117)
  std::vector<float> return3DEdges(unsigned lvl) const { return readF(REVO_PLANE_EDGES3D, lvl, 4); }  // 4 x N col-major
  // imgpyramidrgbd.cpp:279-327: 8 x N column-major (X,Y,Z,1,r,g,b,1)
  std::vector<float> generateColoredPcl(unsigned lvl, bool densePcl) const {
    size_t n = 0;
    check(revo_pyramid_colored_pcl(pyr_, (int)lvl, densePcl ? 1 : 0, nullptr, 0, &n), "generateColoredPcl");
    std::vector<float> v(n * 8);
    if (n) check(revo_pyramid_colored_pcl(pyr_, (int)lvl, densePcl ? 1 : 0, v.data(), n, &n), "generateColoredPcl");
    return v;
  }
  template <class MatX>
  void generateColoredPcl(unsigned lvl, MatX& clrPcl, bool densePcl) const {  // the reference's signature (Eigen::MatrixXf)
    const std::vector<float> v = generateColoredPcl(lvl, densePcl);
    clrPcl.resize(8, (long)(v.size() / 8));
    std::copy(v.begin(), v.end(), clrPcl.data());
  }
  std::vector<float> returnOptimizationStructure(unsigned lvl) const { return readF(REVO_PLANE_GRADTABLE, lvl, 4); }
  std::vector<float> returnDistTransform(unsigned lvl) const { return readF(REVO_PLANE_DT, lvl, 1); }
  std::vector<float> returnDepth(unsigned lvl) const { return readF(REVO_PLANE_DEPTH, lvl, 1); }
  std::vector<uint8_t> returnEdges(unsigned lvl) const { return readU8(REVO_PLANE_EDGES, lvl); }
  std::vector<uint8_t> returnOrigEdges(unsigned lvl) const { return readU8(REVO_PLANE_EDGES_ORIG, lvl); }
  std::vector<uint8_t> returnGray(unsigned lvl) const { return readU8(REVO_PLANE_GRAY, lvl); }
#endif
  std::array<float, 9> returnK(unsigned lvl) const {  // column-major 3x3
    const Camera& c = cameraPyr->at((int)lvl);
    return {c.fx, 0, 0, 0, c.fy, 0, c.cx, c.cy, 1};
  }
  double returnTimestamp() const { return revo_pyramid_timestamp(pyr_); }
  unsigned returnMaxLvl() const { return (unsigned)mSettings.pyr_max_lvl; }
  unsigned returnMinLvl() const { return (unsigned)mSettings.pyr_min_lvl; }

  // pose bookkeeping (imgpyramidrgbd.h:126-151), column-major 4x4
  template <class M4> void setTwf(const M4& T) { std::memcpy(T_w_f, T.data(), sizeof(T_w_f)); }
  void setTwf(const float* T16) { std::memcpy(T_w_f, T16, sizeof(T_w_f)); }
  const float* getTransKFtoWorld() const { return T_w_f; }

  int frameId = 0;
  std::shared_ptr<CameraPyr> cameraPyr;
  revo_pyr* handle() const { return pyr_; }

 private:
  void setIdentity() { std::memset(T_w_f, 0, sizeof(T_w_f)); T_w_f[0] = T_w_f[5] = T_w_f[10] = T_w_f[15] = 1.f; }
  std::vector<float> readF(revo_plane what, unsigned lvl, int k) const {
    size_t n = 0;
    check(revo_pyramid_read(pyr_, what, (int)lvl, nullptr, 0, &n), "accessor");
    std::vector<float> v(n * k);
    if (n) check(revo_pyramid_read(pyr_, what, (int)lvl, v.data(), v.size() * sizeof(float), &n), "accessor");
    return v;
  }
  std::vector<uint8_t> readU8(revo_plane what, unsigned lvl) const {
    size_t n = 0;
    check(revo_pyramid_read(pyr_, what, (int)lvl, nullptr, 0, &n), "accessor");
    std::vector<uint8_t> v(n);
    if (n) check(revo_pyramid_read(pyr_, what, (int)lvl, v.data(), v.size(), &n), "accessor");
    return v;
  }
#if defined(REVO_MATXF) && defined(REVO_VEC4F) && defined(REVO_CVMAT)
  template <class T> using Slots = std::vector<std::unique_ptr<T>>;
  template <class T> static std::unique_ptr<T>& slot(Slots<T>& v, unsigned lvl) {
    if (v.size() <= lvl) v.resize(lvl + 1);
    return v[lvl];
  }
  const REVO_CVMAT& matF(Slots<REVO_CVMAT>& v, revo_plane what, unsigned lvl) const {
    auto& m = slot(v, lvl);
    if (!m) {
      const std::vector<float> d = readF(what, lvl, 1);
      const Camera& c = cameraPyr->at((int)lvl);
      m.reset(new REVO_CVMAT());
      m->create((int)c.height, (int)c.width, CV_32FC1);
      for (size_t y = 0; y < c.height; ++y) std::memcpy(m->data + y * (size_t)m->step, d.data() + y * c.width, c.width * sizeof(float));
    }
    return *m;
  }
  const REVO_CVMAT& matU8(Slots<REVO_CVMAT>& v, revo_plane what, unsigned lvl) const {
    auto& m = slot(v, lvl);
    if (!m) {
      const std::vector<uint8_t> d = readU8(what, lvl);
      const Camera& c = cameraPyr->at((int)lvl);
      m.reset(new REVO_CVMAT());
      m->create((int)c.height, (int)c.width, CV_8UC1);
      for (size_t y = 0; y < c.height; ++y) std::memcpy(m->data + y * (size_t)m->step, d.data() + y * c.width, c.width);
    }
    return *m;
  }
  mutable Slots<REVO_MATXF> m3d_;
  mutable Slots<std::vector<REVO_VEC4F>> tab_;
  mutable Slots<REVO_CVMAT> dt_, depth_, edges_, orig_, gray_;
#endif
  ImgPyramidSettings mSettings;
  revo_pyr* pyr_ = nullptr;
  float T_w_f[16];
};

// ---- Optimizer -------------------------------------------------------------------
class Optimizer {  // optimizer.h:114-186
 public:
  struct ResidualInfo : revo_residual_info {  // optimizer.h:118-140
    ResidualInfo() { clearAll(); }
    void clearAll() { good_pts_edges = bad_pts_edges = 0; sum_error_unweighted = sum_error_weighted = 0.f; }
    int& goodPtsEdges() { return good_pts_edges; }
    int& badPtsEdges() { return bad_pts_edges; }
  };
  Optimizer(const OptimizerSettings& settings, const std::shared_ptr<CameraPyr>& cam) : mSettings(settings), cam_(cam) {}
  explicit Optimizer(const OptimizerSettings& settings) : mSettings(settings) {}  // optimizer.h:166
  void bind(const std::shared_ptr<CameraPyr>& cam) { cam_ = cam; }
  // float trackFrames(refFrame, currFrame, Matrix3f& R, Vector3f& T, int lvl, ResidualInfo&), optimizer.cpp:235-311
  template <class M3, class V3>
  float trackFrames(const std::shared_ptr<ImgPyramidRGBD>& refFrame, const std::shared_ptr<ImgPyramidRGBD>& currFrame, M3& R,
                    V3& T, int lvl, ResidualInfo& resInfo) {
    float err = 0.f;
    if (!cam_) cam_ = refFrame->cameraPyr;
    check(revo_optimizer_track_level(cam_->ctx(), refFrame->handle(), currFrame->handle(), R.data(), T.data(), lvl, &resInfo, &err),
          "Optimizer::trackFrames");
    return err;
  }

 private:
  OptimizerSettings mSettings;
  std::shared_ptr<CameraPyr> cam_;
};

// ---- TrackerNew ------------------------------------------------------------------
class TrackerNew {  // tracker.h:56-105
 public:
  enum TrackerStatus { TRACKER_STATE_OK, TRACKER_STATE_LOST, TRACKER_STATE_NEW_KF, TRACKER_STATE_UNKNOWN };
  int histogramLevel;
  // TrackerNew(const TrackerSettings&, const ImgPyramidSettings&) + the shared CameraPyr (device context)
  TrackerNew(const TrackerSettings& config, const ImgPyramidSettings& pyrConfig, const std::shared_ptr<CameraPyr>& cam)
      : histogramLevel(config.histogram_level), mSettings(config), mPyrConfig(pyrConfig), cam_(cam),
        mOptimizer(config.optimizerSettings, cam) {
    check(revo_ctx_set_tracker(cam->ctx(), &config.optimizerSettings, &config), "TrackerNew");
  }
  // the reference's constructor (tracker.h:71): no camera pyramid -- the device context is the one of the first
  // frame the tracker sees (every ImgPyramidRGBD carries its cameraPyr, imgpyramidrgbd.h:42)
  TrackerNew(const TrackerSettings& config, const ImgPyramidSettings& pyrConfig)
      : histogramLevel(config.histogram_level), mSettings(config), mPyrConfig(pyrConfig), cam_(nullptr),
        mOptimizer(config.optimizerSettings, nullptr) {}
  // tracker.cpp:294-353
  template <class M3, class V3>
  TrackerStatus trackFrames(M3& R, V3& T, float& error, const std::shared_ptr<ImgPyramidRGBD>& refFrame,
                            const std::shared_ptr<ImgPyramidRGBD>& currFrame) {
    int status = TRACKER_STATE_UNKNOWN;
    bind(refFrame->cameraPyr);
    check(revo_tracker_track_frames(cam_->ctx(), refFrame->handle(), currFrame->handle(), R.data(), T.data(), &error, &status,
                                    nullptr, lastEvals),
          "TrackerNew::trackFrames");
    return (TrackerStatus)status;
  }
  // tracker.cpp:118-201
  template <class M4>
  TrackerStatus assessTrackingQuality(const M4& estimatedPose, const std::shared_ptr<ImgPyramidRGBD>& currFrame) {
    int status = TRACKER_STATE_OK;
    bind(currFrame->cameraPyr);
    check(revo_tracker_assess_quality(cam_->ctx(), estimatedPose.data(), currFrame->handle(), &status, nullptr, nullptr),
          "assessTrackingQuality");
    return (TrackerStatus)status;
  }
  // tracker.cpp:209-223: the reference passes frame->return3DEdges(lvl); here the cloud stays in HBM,
  // so the source pyramid and the level are passed instead
  template <class M4>
  void addOldPclAndPose(const std::shared_ptr<ImgPyramidRGBD>& src, int lvl, const M4& worldPose, double timeStamp) {
    bind(src->cameraPyr);
    check(revo_tracker_add_old_pcl(cam_->ctx(), src->handle(), lvl, worldPose.data(), timeStamp), "addOldPclAndPose");
  }
  // the reference's signature, addOldPclAndPose(const Eigen::MatrixXf& pcl, const Eigen::Matrix4f& worldPose, double)
  // (tracker.cpp:209-223; called with frame->return3DEdges(histogramLevel), system.cpp:173,259): a matrix that is the
  // mirror of a pyramid's edge list is copied on the device, any other 4 x N matrix is uploaded
  template <class MatX, class M4, class = decltype(std::declval<const MatX&>().cols())>
  void addOldPclAndPose(const MatX& pcl, const M4& worldPose, double timeStamp) {
    revo_pyr* src = nullptr;
    int lvl = 0;
    if (!cam_) fatal(REVO_ERR_INVALID_ARG, "addOldPclAndPose before the tracker has seen a frame");
    if (MirrorRegistry::get().find(pcl.data(), &src, &lvl))
      check(revo_tracker_add_old_pcl(cam_->ctx(), src, lvl, worldPose.data(), timeStamp), "addOldPclAndPose");
    else
      check(revo_tracker_add_old_pcl_host(cam_->ctx(), pcl.data(), (size_t)pcl.cols(), worldPose.data(), timeStamp), "addOldPclAndPose");
  }
  // binds the tracker to the device context of the frames it is used with
  void bind(const std::shared_ptr<CameraPyr>& cam) {
    if (cam_ == cam) return;
    if (cam_) fatal(REVO_ERR_INVALID_ARG, "TrackerNew used with frames of two camera pyramids");
    cam_ = cam;
    mOptimizer.bind(cam);
    check(revo_ctx_set_tracker(cam->ctx(), &mSettings.optimizerSettings, &mSettings), "TrackerNew");
  }
  void clearUpPastLists() { check(revo_tracker_clear_past(cam_->ctx()), "clearUpPastLists"); }  // tracker.cpp:248-257
  int32_t lastEvals[REVO_MAX_LEVELS] = {0, 0, 0, 0, 0, 0};

 private:
  TrackerSettings mSettings;
  ImgPyramidSettings mPyrConfig;
  std::shared_ptr<CameraPyr> cam_;
  Optimizer mOptimizer;
};

// ---- REVO: the sequencing of REVO::start (system.cpp:84-305) ------------------------
// submit() is the producer side (IOWrapperRGBD::generateImgPyramid: build + queue, asynchronous on
// the device), trackNext() one body of the consumer loop; poses are curr->world, column-major 4x4.
class REVO {
 public:
  explicit REVO(const std::shared_ptr<CameraPyr>& cam) : cam_(cam) { check(revo_vo_create(cam->ctx(), &vo_), "REVO"); }
  ~REVO() { revo_vo_destroy(vo_); }
  REVO(const REVO&) = delete;
  REVO& operator=(const REVO&) = delete;
  void submit(const uint8_t* bgr, size_t bgr_stride, const float* depth_m, size_t depth_stride, double ts) {
    check(revo_vo_submit(vo_, bgr, bgr_stride, depth_m, depth_stride, ts), "REVO::submit");
  }
  template <class MatRGB, class MatDepth>
  void submit(const MatRGB& rgb, const MatDepth& depth, double ts) {
    submit((const uint8_t*)rgb.data, (size_t)rgb.step, (const float*)depth.data, (size_t)depth.step, ts);
  }
  // returns false when the queue is empty
  template <class M4>
  bool trackNext(M4& pose, bool* newKeyframe = nullptr, double* timestamp = nullptr) {
    if (revo_vo_queued(vo_) == 0) return false;
    int kf = 0;
    check(revo_vo_track_next(vo_, pose.data(), &kf, timestamp), "REVO::trackNext");
    if (newKeyframe) *newKeyframe = kf != 0;
    return true;
  }
  int numKeyframes() const { return revo_vo_num_keyframes(vo_); }

  // The reference's threading (system.cpp:96, iowrapperRGBD.cpp:279-288): `produce(*this)` runs on an IO
  // thread and calls submit() for the next frame (returns false when the sequence is over);
  // `consume(pose_colmajor16, newKeyframe, timestamp)` runs on the calling thread for every tracked frame.
  // At most maxQueue pyramids are in flight.  An exception on either side stops both and is rethrown here.
  template <class Produce, class Consume>
  void run(Produce&& produce, Consume&& consume, int maxQueue = 4) {
    check(revo_vo_set_max_queue(vo_, maxQueue), "REVO::run");  // bounded queue inside the library, stream open
    std::mutex m;
    std::exception_ptr err;
    bool stop = false;
    std::thread io([&] {
      try {
        for (;;) {
          { std::lock_guard<std::mutex> lk(m); if (stop) break; }
          if (!produce(*this)) break;  // submit() blocks while maxQueue pyramids wait
        }
      } catch (...) {
        std::lock_guard<std::mutex> lk(m);
        if (!err) err = std::current_exception();
      }
      revo_vo_close(vo_);  // end of stream: revo_vo_wait_frame returns 0 once the queue has drained
    });
    try {
      while (revo_vo_wait_frame(vo_) == 1) {
        std::array<float, 16> pose;
        bool kf = false;
        double ts = 0.0;
        trackNext(pose, &kf, &ts);
        consume(pose, kf, ts);
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(m);
      if (!err) err = std::current_exception();
      stop = true;
    }
    revo_vo_set_max_queue(vo_, 0);  // never leave the IO thread blocked on a full queue
    io.join();
    if (err) std::rethrow_exception(err);
  }

 private:
  std::shared_ptr<CameraPyr> cam_;
  revo_vo* vo_ = nullptr;
};

// ---- tiny column-major helpers for hosts without Eigen ---------------------------
struct Mat3f { float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; float* data() { return m; } const float* data() const { return m; } };
struct Vec3f { float v[3] = {0, 0, 0}; float* data() { return v; } const float* data() const { return v; } };
struct Mat4f {
  float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float* data() { return m; }
  const float* data() const { return m; }
  static Mat4f fromRT(const Mat3f& R, const Vec3f& T) {  // transformFromRT, system.h
    Mat4f o;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) o.m[c * 4 + r] = R.m[c * 3 + r];
    o.m[12] = T.v[0]; o.m[13] = T.v[1]; o.m[14] = T.v[2];
    return o;
  }
  Mat4f operator*(const Mat4f& B) const {
    Mat4f o;
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r)
        o.m[c * 4 + r] = m[r] * B.m[c * 4] + m[4 + r] * B.m[c * 4 + 1] + m[8 + r] * B.m[c * 4 + 2] + m[12 + r] * B.m[c * 4 + 3];
    return o;
  }
  Mat4f inverseRigid() const {  // inverse of a rigid transform
    Mat4f o;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) o.m[c * 4 + r] = m[r * 4 + c];
    for (int r = 0; r < 3; ++r) o.m[12 + r] = -(o.m[r] * m[12] + o.m[4 + r] * m[13] + o.m[8 + r] * m[14]);
    return o;
  }
};

}  // namespace revo
