// Compile-only check that the calls ros_rebvio makes (ros_rebvio/src/ros_rebvio.cpp:15-82, test_ros_rebvio.cpp:17-27,
// rebvio/test/test_rebvio.cpp:7-14) are accepted unchanged by include/rebvio/*.hpp.
#include <cmath>
#include <functional>
#include <string>

#include "rebvio/rebvio.hpp"
#include "rebvio/scale_space.hpp"
#include "rebvio/util/timer.hpp"

namespace rebvio {
struct RosRebvioConfigLike {
  std::string imu_topic, cam_topic;
  rebvio::RebvioConfig config;
};
}  // namespace rebvio

int callers(rebvio::RosRebvioConfigLike _config) {
  rebvio::Rebvio rebvio_(_config.config);  // ros_rebvio.cpp:15
  int px = 0;
  std::function<void(rebvio::types::Odometry & _odometry)> odometryCallback = [&](rebvio::types::Odometry& _odometry) {
    double rot[3] = {_odometry.orientation[0], _odometry.orientation[1], _odometry.orientation[2]};
    double pos[3] = {_odometry.position[0], _odometry.position[1], _odometry.position[2]};
    unsigned long long ns = _odometry.ts_us * 1000;
    (void)rot; (void)pos; (void)ns;
  };
  rebvio_.registerOdometryCallback(odometryCallback);  // :30
  std::function<void(cv::Mat & _edge_image, rebvio::EdgeMap::SharedPtr & _map)> edgeImageCallback =
      [&](cv::Mat& _edge_image, rebvio::EdgeMap::SharedPtr& _map) {
        if (_edge_image.type() != CV_8UC1) {}
        if (_map) {
          for (int i = 0; i < _map->size(); ++i) {  // :43-46
            px += (int)std::round((*_map)[i].pos[1]) + (int)std::round((*_map)[i].pos[0]);
          }
        }
      };
  rebvio_.registerEdgeImageCallback(edgeImageCallback);  // :51
  cv::Mat image(480, 752, CV_8UC1);
  rebvio_.imageCallback(rebvio::types::Image{(uint64_t)123456789 / 1000, image});  // :70
  rebvio_.imuCallback(rebvio::types::Imu{(uint64_t)1000, TooN::makeVector((rebvio::types::Float)0.1, (rebvio::types::Float)0.2, (rebvio::types::Float)0.3),
                                         TooN::makeVector((rebvio::types::Float)0.0, (rebvio::types::Float)9.8, (rebvio::types::Float)0.0)});  // :80-82
  // rebvio/test/test_rebvio.cpp:7-14
  rebvio::Core core(std::make_shared<rebvio::Camera>());
  rebvio::types::Vector3f Vgv = TooN::makeVector(-4.06833e-05f, 9.40667e-05f, 5.70767e-05f);
  rebvio::types::Float dt = 0.05;
  rebvio::types::Vector3f Av = TooN::makeVector(0.0f, 0.0f, 0.0f);
  rebvio::types::Matrix3f R = TooN::Data(1, 8.83134e-05, -7.48149e-05, -8.831e-05, 1, 4.57494e-05, 7.4819e-05, -4.57428e-05, 1);
  core.estimateLs4Acceleration(-Vgv / dt, Av, R, dt);
  return px;
}

// The rest of the library's public surface (a user of rebvio links against these although ros_rebvio does not call them):
// core.hpp:20-79,122,138,187; edge_map.hpp:93-94; scale_space.hpp:22-96; util/timer.hpp:18-32.
int library_users(rebvio::Camera::SharedPtr camera, rebvio::EdgeMap::SharedPtr map, rebvio::EdgeMap::SharedPtr other) {
  REBVIO_TIMER_TICK();
  rebvio::DistanceField field(480, 752, 40.0);
  field.build(map);
  rebvio::DistanceFieldElement& e = field[0];
  int acc = e.id + e.distance + field.map()->size();
  rebvio::Core core(camera);
  rebvio::types::KeyLine& k = (*map)[0];
  rebvio::types::Float dx, dy, fi = 0;
  int mnum = 0;
  acc += rebvio::Core::testfk(k, (*other)[0], 0.5f) ? 1 : 0;
  acc += (int)core.calculatefJ(map, 0, dx, dy, k, k.pos[0], k.pos[1], mnum, fi);
  rebvio::types::Vector3f v = TooN::makeVector(0.0f, 0.0f, 0.0f);
  core.updateInverseDepthARLU(k, v);
  rebvio::types::Matrix3f I = TooN::Data(1, 0, 0, 0, 1, 0, 0, 0, 1);
  acc += other->searchMatch(k, v, I, I, 40.0f);
  rebvio::ScaleSpace space(camera);
  cv::Mat img(480, 752, CV_32FC1);
  space.build(img);
  acc += space.dog().rows + space.mag().cols;
  rebvio::FastGaussian fg(camera, 3.56359f, 3);
  acc += fg.smooth(img).rows + fg.n_ + fg.widths_[0] + (int)fg.sigma_true_ + fg.divisors_[0].rows;
  REBVIO_TIMER_TOCK();
  return acc;
}
