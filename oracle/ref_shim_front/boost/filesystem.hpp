// TEST INFRASTRUCTURE: the boost::filesystem calls of the reference's front ends (path / wpath, exists, create_directory,
// filename) on top of the stand-in the library build uses
#ifndef L3D_REF_SHIM_FRONT_BOOST_FS_
#define L3D_REF_SHIM_FRONT_BOOST_FS_
#include <ostream>
#include <string>
#include <sys/stat.h>
namespace boost { namespace filesystem {
class path {
public:
    path() {}
    path(const std::string& s) : s_(s) {}
    path(const char* s) : s_(s) {}
    const std::string& string() const { return s_; }
    const char* c_str() const { return s_.c_str(); }
    path parent_path() const { const size_t p = s_.find_last_of("/\\"); return path(p == std::string::npos ? std::string() : s_.substr(0, p)); }
    path filename() const { const size_t p = s_.find_last_of("/\\"); return path(p == std::string::npos ? s_ : s_.substr(p + 1)); }
private:
    std::string s_;
};
typedef path wpath;
inline std::ostream& operator<<(std::ostream& o, const path& p) { return o << '"' << p.string() << '"'; }
inline bool exists(const path& p) { struct stat st; return ::stat(p.string().c_str(), &st) == 0; }
inline bool create_directory(const path& p) { return ::mkdir(p.string().c_str(), 0755) == 0; }
}}
#endif
