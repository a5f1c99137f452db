// TEST INFRASTRUCTURE: stand-in for the three boost::filesystem calls the reference makes
#ifndef L3D_REF_SHIM_BOOST_FS_
#define L3D_REF_SHIM_BOOST_FS_
#include <string>
#include <sys/stat.h>
namespace boost { namespace filesystem {
class path { public: path() {} path(const std::string& s) : s_(s) {} path(const char* s) : s_(s) {} const std::string& string() const { return s_; } private: std::string s_; };
inline bool exists(const path& p) { struct stat st; return ::stat(p.string().c_str(), &st) == 0; }
inline bool create_directory(const path& p) { return ::mkdir(p.string().c_str(), 0755) == 0; }
}}
#endif
