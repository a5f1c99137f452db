// TEST INFRASTRUCTURE: no-op stand-ins for Boost.Serialization (segment caches / .bin output are never
// exercised by the oracle driver; the templates only have to compile)
#ifndef L3D_REF_SHIM_BOOST_SER_
#define L3D_REF_SHIM_BOOST_SER_
#include <cstddef>
#include <iosfwd>
namespace boost { namespace serialization {
class access {};
struct nothing {};
template <class T> nothing make_nvp(const char*, T&) { return nothing(); }
template <class T> nothing make_nvp(const char*, const T&) { return nothing(); }
template <class T> nothing make_array(T*, std::size_t) { return nothing(); }
template <class T> nothing make_array(const T*, std::size_t) { return nothing(); }
}
namespace archive {
struct bool_true { static const bool value = true; };
struct bool_false { static const bool value = false; };
class binary_oarchive { public: typedef bool_false is_loading; typedef bool_true is_saving; binary_oarchive(std::ostream&) {}
    template <class T> binary_oarchive& operator&(const T&) { return *this; } template <class T> binary_oarchive& operator<<(const T&) { return *this; } };
class binary_iarchive { public: typedef bool_true is_loading; typedef bool_false is_saving; binary_iarchive(std::istream&) {}
    template <class T> binary_iarchive& operator&(const T&) { return *this; } template <class T> binary_iarchive& operator>>(const T&) { return *this; } };
}}
#endif
