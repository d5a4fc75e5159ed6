// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
// Minimal stand-in for <boost/thread.hpp>: the reference's open_karto uses exactly three
// Boost names (lesson6/lib/open_karto/include/open_karto/Karto.h:37,5195,5245-5343):
// boost::shared_mutex, boost::shared_lock, boost::unique_lock.  Boost is not installed in
// this image, so we alias them to their C++14 std equivalents.  Nothing from the reference
// is copied here.
#pragma once
#include <mutex>
#include <shared_mutex>
namespace boost {
using shared_mutex = std::shared_timed_mutex;
template <class M> using shared_lock = std::shared_lock<M>;
template <class M> using unique_lock = std::unique_lock<M>;
}  // namespace boost
