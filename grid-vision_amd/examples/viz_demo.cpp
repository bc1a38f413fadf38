// viz_demo.cpp -- the marker and overlay builders of grid_vision/viz_specs.hpp on a hand-made tick result, printed one
// line per marker / rectangle; host code only (no GPU, no ROS, no OpenCV):
//   g++ -std=c++17 -O2 viz_demo.cpp -o viz_demo        (header-only: nothing to link)
// tests/test_host_side.py compares the lines with hand-derived known answers.
#include <cstdio>
#include <vector>

#include "../include/grid_vision/viz_specs.hpp"

int main()
{
  using namespace grid_vision;
  // static boxes: red light, 60 sign, an UNKNOWN class (no marker), green light, orange light, 30 sign
  const std::vector<BoundingBox> st = {{420, 100, 470, 160, 0.8f, 5},  {40, 60, 100, 120, 0.7f, 7}, {1, 2, 3, 4, 0.6f, 10},
                                       {10, 20, 30, 40, 0.65f, 3},     {50, 60, 70, 80, 0.64f, 4},  {5, 6, 70, 80, 0.63f, 6}};
  const std::vector<geometry::Point> pts = {{12.5, -3.25, 4.0}, {20.0, 5.5, 2.25}, {1, 1, 1}, {7.0, 8.0, 9.0}, {-1.5, 2.5, 3.5}, {30.0, -6.0, 1.0}};
  std::vector<LShapePose> boxes(2);
  boxes[0] = LShapePose{10.0, 2.0, 0.5, 0.0, 0.0, 0.38268343236508978, 0.92387953251128674, 4.5, 1.8, 1.6};   // vision branch: all dims
  boxes[1] = LShapePose{25.0, -4.0, 0.25, 0.0, 0.1, 0.0, 0.99, 3.2, 1.1, 0.0};                                  // PCA branch: height never set
  for (const MarkerSpec &m : buildObjectVisualizations(boxes, pts, st, "hero"))
    std::printf("marker id %d ns %s type %d action %d life %.3f frame %s pos %.6f %.6f %.6f quat %.9f %.9f %.9f %.9f scale %.3f %.3f %.3f rgba %.2f %.2f %.2f %.2f text [%s]\n",
                m.id, m.ns.c_str(), m.type, m.action, m.lifetime_s, m.frame_id.c_str(), m.px, m.py, m.pz, m.qx, m.qy, m.qz, m.qw, m.sx, m.sy,
                m.sz, m.r, m.g, m.b, m.a, m.text.c_str());
  // overlay: fractional corners (what a caller other than denormalize could hand over), truncation towards zero
  const std::vector<BoundingBox> all = {{100.9, 50.2, 220.7, 300.5, 0.95f, 9}, {0.0, 3.0, 639.0, 479.0, 0.6f, 2}, {330, 200, 460, 330, 0.123456f, 42}};
  const std::vector<OverlaySpec> ov = buildDetectionOverlay(all);
  for (const OverlaySpec &o : ov)
    std::printf("overlay rect %d %d %d %d text_at %d %d label [%s] rgb %d %d %d thickness %d %d font %.2f\n", o.x, o.y, o.w, o.h, o.text_x,
                o.text_y, o.label.c_str(), o.r, o.g, o.b, o.box_thickness, o.text_thickness, o.font_scale);
  // the rectangle of the first box drawn into a 640 x 480 rgb8 image: count and corners of the green pixels
  std::vector<uint8_t> img(640 * 480 * 3, 7);
  drawOverlayRectangles(img.data(), 640, 480, {ov[0]});
  long green = 0;
  int minx = 640, miny = 480, maxx = -1, maxy = -1;
  for (int y = 0; y < 480; ++y)
    for (int x = 0; x < 640; ++x) {
      const uint8_t *p = &img[(size_t)(y * 640 + x) * 3];
      if (p[0] == 0 && p[1] == 255 && p[2] == 0) {
        ++green;
        if (x < minx) minx = x;
        if (x > maxx) maxx = x;
        if (y < miny) miny = y;
        if (y > maxy) maxy = y;
      }
    }
  std::printf("drawn green %ld bbox %d %d %d %d\n", green, minx, miny, maxx, maxy);
  return 0;
}
