// TEST INFRASTRUCTURE ONLY -- stand-in for Iridescence's guik::LightViewer: invoke() runs the task at once
// (the real viewer queues it for its render thread).
#pragma once
#include <string>

#include <glk/pointcloud_buffer.hpp>

namespace guik {
class LightViewer {
public:
  static LightViewer* instance() {
    static LightViewer v;
    return &v;
  }
  template <typename F>
  void invoke(const F& f) { f(); }
  void append_text(const std::string&) {}
};
}  // namespace guik
