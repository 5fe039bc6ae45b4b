// Minimal stand-in for the subset of google-glog that pbrt-v3 uses
// (LOG/VLOG/CHECK*/DCHECK*, a handful of FLAGS_ globals). Written for this
// repo's oracle build only: the reference's glog submodule is an empty
// directory in /root/reference. Test infrastructure.
#ifndef PBRT_ORACLE_STUB_GLOG_LOGGING_H
#define PBRT_ORACLE_STUB_GLOG_LOGGING_H
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <iostream>
#include <sstream>
#include <string>

namespace google {
inline void InitGoogleLogging(const char *) {}
struct NullStream {
    template <typename T> NullStream &operator<<(const T &) { return *this; }
    NullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
struct FatalStream {
    std::ostringstream ss;
    FatalStream(const char *file, int line, const char *what) {
        ss << file << ":" << line << " " << what << " ";
    }
    template <typename T> FatalStream &operator<<(const T &v) { ss << v; return *this; }
    FatalStream &operator<<(std::ostream &(*f)(std::ostream &)) { ss << f; return *this; }
    ~FatalStream() {
        std::fprintf(stderr, "FATAL %s\n", ss.str().c_str());
        std::fflush(stderr);
        std::abort();
    }
};
struct Voidify {
    void operator&(const NullStream &) {}
    void operator&(const FatalStream &) {}
};
}  // namespace google

extern int FLAGS_v;
extern int FLAGS_minloglevel;
extern int FLAGS_stderrthreshold;
extern bool FLAGS_logtostderr;
extern std::string FLAGS_log_dir;

#define PBRT_STUB_NULL_ (true) ? (void)0 : google::Voidify() & google::NullStream()
#define PBRT_STUB_FATAL_IF_(cond, what) \
    (!(cond)) ? (void)0 : google::Voidify() & google::FatalStream(__FILE__, __LINE__, what)

#define LOG_INFO_ PBRT_STUB_NULL_
#define LOG_WARNING_ PBRT_STUB_NULL_
#define LOG_ERROR_ PBRT_STUB_NULL_
#define LOG_FATAL_ PBRT_STUB_FATAL_IF_(true, "LOG(FATAL)")
#define LOG(sev) LOG_##sev##_
#define VLOG(n) PBRT_STUB_NULL_

#define CHECK(c) PBRT_STUB_FATAL_IF_(!(c), "Check failed: " #c)
#define CHECK_EQ(a, b) PBRT_STUB_FATAL_IF_(!((a) == (b)), "Check failed: " #a " == " #b)
#define CHECK_NE(a, b) PBRT_STUB_FATAL_IF_(!((a) != (b)), "Check failed: " #a " != " #b)
#define CHECK_LT(a, b) PBRT_STUB_FATAL_IF_(!((a) < (b)), "Check failed: " #a " < " #b)
#define CHECK_LE(a, b) PBRT_STUB_FATAL_IF_(!((a) <= (b)), "Check failed: " #a " <= " #b)
#define CHECK_GT(a, b) PBRT_STUB_FATAL_IF_(!((a) > (b)), "Check failed: " #a " > " #b)
#define CHECK_GE(a, b) PBRT_STUB_FATAL_IF_(!((a) >= (b)), "Check failed: " #a " >= " #b)
#define CHECK_NEAR(a, b, eps) \
    PBRT_STUB_FATAL_IF_(!(std::abs((a) - (b)) <= (eps)), "Check failed: " #a " near " #b)

#ifdef NDEBUG
#define DCHECK(c) PBRT_STUB_NULL_
#define DCHECK_EQ(a, b) PBRT_STUB_NULL_
#define DCHECK_NE(a, b) PBRT_STUB_NULL_
#define DCHECK_LT(a, b) PBRT_STUB_NULL_
#define DCHECK_LE(a, b) PBRT_STUB_NULL_
#define DCHECK_GT(a, b) PBRT_STUB_NULL_
#define DCHECK_GE(a, b) PBRT_STUB_NULL_
#else
#define DCHECK(c) CHECK(c)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK_NE(a, b) CHECK_NE(a, b)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
#endif
#endif
