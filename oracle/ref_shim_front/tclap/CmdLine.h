// TEST INFRASTRUCTURE: the part of TCLAP the reference's front ends use (CmdLine, ValueArg<T>, "-f value" / "--name value")
#ifndef L3D_REF_SHIM_FRONT_TCLAP_H_
#define L3D_REF_SHIM_FRONT_TCLAP_H_
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace TCLAP {
class Arg {
public:
    Arg(const std::string& f, const std::string& n, bool req) : flag_(f), name_(n), req_(req), set_(false) {}
    virtual ~Arg() {}
    virtual void assign(const std::string& v) = 0;
    std::string flag_, name_;
    bool req_, set_;
};
template <class T> class ValueArg : public Arg {
public:
    ValueArg(const std::string& flag, const std::string& name, const std::string&, bool req, T def, const std::string&)
        : Arg(flag, name, req), v_(def) {}
    T& getValue() { return v_; }
    void assign(const std::string& s) { std::istringstream is(s); is >> v_; set_ = true; }
private:
    T v_;
};
template <> inline void ValueArg<std::string>::assign(const std::string& s) { v_ = s; set_ = true; }
class CmdLine {
public:
    explicit CmdLine(const std::string&, char = ' ', const std::string& = "none", bool = true) {}
    void add(Arg& a) { args_.push_back(&a); }
    void parse(int argc, char** argv) {
        for (int i = 1; i + 1 < argc; i += 2) {
            const std::string key = argv[i];
            bool found = false;
            for (size_t k = 0; k < args_.size(); ++k)
                if (key == "-" + args_[k]->flag_ || key == "--" + args_[k]->name_) { args_[k]->assign(argv[i + 1]); found = true; }
            if (!found) throw std::runtime_error("unknown argument " + key);
        }
        for (size_t k = 0; k < args_.size(); ++k)
            if (args_[k]->req_ && !args_[k]->set_) throw std::runtime_error("required argument missing: " + args_[k]->name_);
    }
private:
    std::vector<Arg*> args_;
};
}  // namespace TCLAP
#endif
