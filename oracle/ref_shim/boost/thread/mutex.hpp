// TEST INFRASTRUCTURE: stand-in for boost::mutex (Boost is not installed in this image)
#ifndef L3D_REF_SHIM_BOOST_MUTEX_
#define L3D_REF_SHIM_BOOST_MUTEX_
#include <mutex>
namespace boost { class mutex { public: void lock() { m_.lock(); } void unlock() { m_.unlock(); } private: std::mutex m_; }; }
#endif
