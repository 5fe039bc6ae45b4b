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
}
#endif
