// Error()/Warning() with pbrt's "print once, continue" behaviour
// (src/core/error.cpp:62-102).  Messages carry the current parser location.
#ifndef PBRT_AMD_HOST_ERROR_H
#define PBRT_AMD_HOST_ERROR_H
#include <string>
namespace pbrt {
struct FileLoc { std::string filename; int line = 0, column = 0; };
extern FileLoc *parserLoc;       // set by the parser while a file is read
extern bool quietWarnings;       // Options::quiet
int ErrorCount();                // number of distinct Error() messages so far
void Warning(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void Error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
// Where the reference stops the process (LOG(FATAL), exit(1) in its parser), this library unwinds to its entry point
// instead: the CLI turns it into exit status 1, the C entry points of include/pbrt_host.h into a null scene -- a malformed
// scene string must not take down a host process (a Python rank of a distributed job) that merely asked for a parse.
struct FatalError {};
[[noreturn]] void Fatal();
}
#endif
