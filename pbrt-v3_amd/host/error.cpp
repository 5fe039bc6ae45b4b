#include "error.h"
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <set>
namespace pbrt {
FileLoc *parserLoc = nullptr;
bool quietWarnings = false;
static int nErrors = 0;
int ErrorCount() { return nErrors; }
// error.cpp:62-89: prefix with file:line:col, print each distinct message once.
static void processError(const char *kind, const char *fmt, va_list args, bool isError) {
    char buf[2048];
    vsnprintf(buf, sizeof(buf), fmt, args);
    std::string msg;
    if (parserLoc) {
        char loc[512];
        snprintf(loc, sizeof(loc), "%s:%d:%d: ", parserLoc->filename.c_str(), parserLoc->line, parserLoc->column);
        msg = loc;
    }
    msg += buf;
    static std::mutex mu;
    static std::set<std::string> seen;
    std::lock_guard<std::mutex> lock(mu);
    if (!seen.insert(msg).second) return;
    if (isError) ++nErrors;
    fprintf(stderr, "%s: %s\n", kind, msg.c_str());
}
void Warning(const char *fmt, ...) {
    if (quietWarnings) return;
    va_list a; va_start(a, fmt); processError("Warning", fmt, a, false); va_end(a);
}
void Error(const char *fmt, ...) {
    va_list a; va_start(a, fmt); processError("Error", fmt, a, true); va_end(a);
}
void Fatal() { throw FatalError(); }
}
