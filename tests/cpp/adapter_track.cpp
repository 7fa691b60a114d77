// Drives the hot path through the C++ adapters exactly like system.cpp drives the
// reference classes (system.cpp:151-175,186-199): first frame -> keyframe, second frame ->
// trackFrames + assessTrackingQuality.  Inputs are raw files written by the pytest side;
// prints the pose so the test can compare it with the Python/C-ABI path bit for bit.
#define REVO_ADAPTERS_THROW
#include <cstdio>
#include <fstream>
#include <vector>

#include "../../revo_amd/cpp/revo_adapters.hpp"

static std::vector<char> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: %s w h ref.bgr ref.depth cur.bgr cur.depth\n", argv[0]); return 2; }
  const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
  revo::ImgPyramidSettings ps;
  const float sx = w / 640.0f, sy = h / 480.0f;
  ps.width = w; ps.height = h; ps.fx *= sx; ps.fy *= sy; ps.cx *= sx; ps.cy *= sy;
  if (w != 640) { ps.hist_patch[0] = 10; ps.hist_patch[1] = 5; ps.hist_patch[2] = 0; }
  revo::TrackerSettings ts;
  auto camPyr = std::make_shared<revo::CameraPyr>(ps);
  revo::TrackerNew tracker(ts, ps, camPyr);
  auto rb = slurp(argv[3]), rd = slurp(argv[4]), cb = slurp(argv[5]), cd = slurp(argv[6]);
  if ((int)rb.size() != w * h * 3 || (int)rd.size() != w * h * 4) { std::fprintf(stderr, "bad input size\n"); return 2; }
  auto kfPyr = std::make_shared<revo::ImgPyramidRGBD>(ps, camPyr, (const uint8_t*)rb.data(), (size_t)w * 3,
                                                      (const float*)rd.data(), (size_t)w * 4, 0.0);
  kfPyr->makeKeyframe();
  revo::Mat4f I;
  kfPyr->setTwf(I);
  tracker.addOldPclAndPose(kfPyr, tracker.histogramLevel, I, kfPyr->returnTimestamp());
  auto currPyr = std::make_shared<revo::ImgPyramidRGBD>(ps, camPyr, (const uint8_t*)cb.data(), (size_t)w * 3,
                                                        (const float*)cd.data(), (size_t)w * 4, 1.0 / 30);
  revo::Mat3f R;
  revo::Vec3f T;
  float error = 0.f;
  tracker.trackFrames(R, T, error, kfPyr, currPyr);
  const revo::Mat4f T_KF_N = revo::Mat4f::fromRT(R, T);
  revo::Mat4f Twkf;
  std::memcpy(Twkf.m, kfPyr->getTransKFtoWorld(), sizeof(Twkf.m));
  const revo::Mat4f currPoseInWorld = Twkf * T_KF_N;
  const int status = tracker.assessTrackingQuality(currPoseInWorld, currPyr);
  std::printf("R");
  for (int i = 0; i < 9; ++i) std::printf(" %.9g", R.m[i]);
  std::printf("\nT %.9g %.9g %.9g\nerr %.9g\nstatus %d\nn0 %zu\n", T.v[0], T.v[1], T.v[2], error, status,
              currPyr->return3DEdges(0).size() / 4);
  // generateColoredPcl(lvl, clrPcl, dense), imgpyramidrgbd.cpp:279-327: count and a checksum of the floats
  const std::vector<float> pcl = kfPyr->generateColoredPcl(1, true);
  double sum = 0;
  for (float v : pcl) sum += v;
  std::printf("pcl %zu %.9g\n", pcl.size() / 8, sum);
  // REVO::start sequencing with the reference's IO thread: 8 frames alternating the two inputs
  {
    revo::REVO vo(camPyr);
    int fed = 0;
    vo.run(
        [&](revo::REVO& r) {
          if (fed == 8) return false;
          const bool odd = (fed & 1) != 0;
          r.submit((const uint8_t*)(odd ? cb.data() : rb.data()), (size_t)w * 3, (const float*)(odd ? cd.data() : rd.data()), (size_t)w * 4,
                   fed / 30.0);
          ++fed;
          return true;
        },
        [&](const std::array<float, 16>& pose, bool kf, double ts) {
          std::printf("vo %d %.6f", kf ? 1 : 0, ts);
          for (int i = 0; i < 16; ++i) std::printf(" %.9g", pose[i]);
          std::printf("\n");
        });
    std::printf("vokf %d\n", vo.numKeyframes());
  }
  // error behaviour: tracking against a non-keyframe must fail like "optimizationStructure not built!"
  try {
    tracker.trackFrames(R, T, error, currPyr, kfPyr);
    std::printf("notkf no-error\n");
  } catch (const std::exception& e) {
    std::printf("notkf error\n");
  }
  return 0;
}
