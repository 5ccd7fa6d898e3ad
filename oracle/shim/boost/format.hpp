// TEST INFRASTRUCTURE ONLY -- stand-in for <boost/format.hpp>: the log lines of
// visual_camera_calibration.cpp are formatted into text nobody reads here.
#pragma once
#include <ostream>
#include <sstream>
#include <string>

namespace boost {
class format {
public:
  explicit format(const std::string& f) : text(f) {}
  template <typename T>
  format& operator%(const T&) {
    return *this;
  }
  std::string text;
};
inline std::ostream& operator<<(std::ostream& os, const format& f) { return os << f.text; }
}  // namespace boost
