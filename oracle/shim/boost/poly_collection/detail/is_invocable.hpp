// TEST INFRASTRUCTURE ONLY -- stand-in for the Boost header include/vlcal/common/frame_traits.hpp pulls in
// (Boost is not installed here): boost::poly_collection::detail::is_invocable == std::is_invocable.
#pragma once
#include <type_traits>

namespace boost {
namespace poly_collection {
namespace detail {
template <typename F, typename... Args>
struct is_invocable : std::is_invocable<F, Args...> {};
}  // namespace detail
}  // namespace poly_collection
}  // namespace boost
