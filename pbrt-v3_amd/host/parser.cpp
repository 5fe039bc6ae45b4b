// .pbrt scene-file tokenizer + directive dispatch (the reference's front end:
// src/core/parser.cpp:786-1090).  Same grammar: quoted strings, [ ] arrays,
// '#' comments, "type name" parameter declarations; numbers go through
// strtol/strtof exactly as parser.cpp:322-372 so float literals round the
// same way.  Directives this build does not implement are reported with
// Error() and skipped, pbrt-style (no exceptions).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <memory>
#include <chrono>
#include <sstream>
#include <thread>
#include <vector>
#include <algorithm>
#include "api.h"
#include "error.h"
#include "paramset.h"

namespace pbrt {
namespace {

struct Tokenizer {
    std::string contents;
    size_t pos = 0;
    FileLoc loc;
    static std::unique_ptr<Tokenizer> FromFile(const std::string &fn) {
        std::ifstream f(fn, std::ios::binary | std::ios::ate);
        if (!f) { Error("%s: unable to open file", fn.c_str()); return nullptr; }
        std::unique_ptr<Tokenizer> t(new Tokenizer);
        const std::streamoff size = f.tellg();  // one read into the final buffer: mesh files run to hundreds of megabytes
        if (size > 0) {
            t->contents.resize((size_t)size);
            f.seekg(0);
            f.read(&t->contents[0], size);
            if (!f) { Error("%s: read error", fn.c_str()); return nullptr; }
        }
        t->loc.filename = fn; t->loc.line = 1; t->loc.column = 0;
        return t;
    }
    static std::unique_ptr<Tokenizer> FromString(const std::string &s) {
        std::unique_ptr<Tokenizer> t(new Tokenizer);
        t->contents = s; t->loc.filename = "<string>"; t->loc.line = 1;
        return t;
    }
    int get() {
        if (pos >= contents.size()) return EOF;
        int ch = (unsigned char)contents[pos++];
        if (ch == '\n') { ++loc.line; loc.column = 0; } else ++loc.column;
        return ch;
    }
    void unget() { --pos; if (contents[pos] == '\n') --loc.line; else --loc.column; }
    // Returns "" at EOF.  Quoted strings keep their quotes (parser.cpp Tokenizer::Next).
    std::string Next() {
        while (true) {
            size_t start = pos;
            int ch = get();
            if (ch == EOF) return "";
            if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r') continue;
            if (ch == '"') {
                bool haveEscaped = false;
                while ((ch = get()) != '"') {
                    if (ch == EOF) { Error("premature EOF"); return ""; }
                    if (ch == '\n') { Error("unterminated string"); return ""; }
                    if (ch == '\\') { haveEscaped = true; if (get() == EOF) { Error("premature EOF"); return ""; } }
                }
                std::string s = contents.substr(start, pos - start);
                if (!haveEscaped) return s;
                std::string out;
                for (size_t i = 0; i < s.size(); ++i) {
                    if (s[i] != '\\') { out += s[i]; continue; }
                    ++i;
                    switch (s[i]) {
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break; case 'r': out += '\r'; break;
                    case 't': out += '\t'; break; case '\\': out += '\\'; break;
                    case '\'': out += '\''; break; case '"': out += '"'; break;
                    default: Error("unexpected escaped character \"%c\"", s[i]); return "";
                    }
                }
                return out;
            }
            if (ch == '[' || ch == ']') return std::string(1, (char)ch);
            if (ch == '#') {
                while ((ch = get()) != EOF) if (ch == '\n' || ch == '\r') { unget(); break; }
                continue;  // comments are swallowed (no --cat mode)
            }
            while ((ch = get()) != EOF) {
                if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '"' || ch == '[' || ch == ']') { unget(); break; }
            }
            return contents.substr(start, pos - start);
        }
    }
    // The body of a bracketed array when it holds nothing but numbers -- the vertex and index lists of a mesh, hundreds of
    // megabytes of text for the scenes this build is meant for.  Called right after the '[': if only number characters and
    // white space lie before the matching ']', the span is cut at white space into one piece per thread, every piece is
    // converted with the very calls the token-by-token path makes (parseNumber: strtol for digit strings, strtof otherwise, so
    // every value rounds as in the reference), `pos` moves behind the ']' and true is returned.  Anything else (strings,
    // comments, booleans, a malformed number) leaves the tokenizer untouched for the general path.
    bool NumberArray(std::vector<double> *out) {
        const char *base = contents.data();
        const size_t n = contents.size();
        size_t end = pos;
        while (end < n && base[end] != ']') {
            const unsigned char c = (unsigned char)base[end];
            if (!((c >= '0' && c <= '9') || c == '.' || c == '-' || c == '+' || c == 'e' || c == 'E' || c == ' ' || c == '\n' || c == '\t' || c == '\r')) return false;
            ++end;
        }
        if (end >= n || end - pos < (size_t)1 << 16) return false;  // short arrays: not worth the threads
        int nThreads = PbrtOptions.nThreads > 0 ? PbrtOptions.nThreads : (int)std::thread::hardware_concurrency();
        nThreads = std::max(1, std::min(nThreads, 64));
        auto isSpace = [](char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; };
        std::vector<size_t> cut((size_t)nThreads + 1);
        for (int t = 0; t <= nThreads; ++t) {
            size_t c = pos + (end - pos) * (size_t)t / (size_t)nThreads;
            while (c > pos && c < end && !isSpace(base[c])) ++c;  // pieces begin at white space (or at the array's ends)
            cut[t] = c;
        }
        std::vector<std::vector<double>> part((size_t)nThreads);
        std::vector<int> bad((size_t)nThreads, 0);
        auto work = [&](int t) {
            std::vector<double> &v = part[t];
            v.reserve((cut[t + 1] - cut[t]) / 6 + 16);
            const char *p = base + cut[t], *e = base + cut[t + 1];
            while (p < e) {
                if (isSpace(*p)) { ++p; continue; }
                const char *q = p;
                bool digits = true;
                while (q < e && !isSpace(*q)) { if (*q < '0' || *q > '9') digits = false; ++q; }
                char *stop = nullptr;
                double val;
                if (q - p == 1 && digits) val = *p - '0';
                else if (digits) val = (double)strtol(p, &stop, 10);
                else val = strtof(p, &stop);  // the token ends at white space or at the ']', which ends the conversion too
                if (stop && stop != q) { bad[t] = 1; return; }  // not one well-formed number: let the general path report it
                v.push_back(val);
                p = q;
            }
        };
        std::vector<std::thread> threads;
        for (int t = 1; t < nThreads; ++t) threads.emplace_back(work, t);
        work(0);
        for (auto &th : threads) th.join();
        for (int t = 0; t < nThreads; ++t) if (bad[t]) return false;
        size_t total = 0;
        for (auto &v : part) total += v.size();
        out->reserve(out->size() + total);
        for (auto &v : part) out->insert(out->end(), v.begin(), v.end());
        for (size_t i = pos; i < end; ++i) if (base[i] == '\n') { ++loc.line; loc.column = 0; }
        pos = end + 1;  // behind the ']'
        if (getenv("PBRT_HOST_TIMING")) fprintf(stderr, "pbrt host: %zu numbers of a %.0f MB array converted on %d threads\n", total, (double)(end - (size_t)(base + cut[0] - base)) / 1e6, nThreads);
        return true;
    }
};

bool isQuoted(const std::string &s) { return s.size() >= 2 && s.front() == '"' && s.back() == '"'; }
std::string dequote(const std::string &s) {
    if (!isQuoted(s)) { Error("\"%s\": expected quoted string", s.c_str()); Fatal(); }
    return s.substr(1, s.size() - 2);
}

double parseNumber(const std::string &str) {  // parser.cpp:322-372
    if (str.size() == 1) {
        if (!(str[0] >= '0' && str[0] <= '9')) { Error("\"%c\": expected a number", str[0]); Fatal(); }
        return str[0] - '0';
    }
    bool isInt = true;
    for (char ch : str) if (!(ch >= '0' && ch <= '9')) isInt = false;
    char *endptr = nullptr;
    double val;
    if (isInt) val = double(strtol(str.c_str(), &endptr, 10));
    else val = strtof(str.c_str(), &endptr);
    if (val == 0 && endptr == str.c_str()) { Error("%s: expected a number", str.c_str()); Fatal(); }
    return val;
}

struct Parser {
    std::vector<std::unique_ptr<Tokenizer>> fileStack;
    bool ungetSet = false;
    std::string ungetValue;

    std::string nextToken(bool required) {
        if (ungetSet) { ungetSet = false; return ungetValue; }
        while (true) {
            if (fileStack.empty()) {
                if (required) { Error("premature EOF"); Fatal(); }
                parserLoc = nullptr;
                return "";
            }
            std::string tok = fileStack.back()->Next();
            if (tok.empty()) {
                fileStack.pop_back();  // (the location the error routines print lived in the tokenizer just destroyed)
                parserLoc = fileStack.empty() ? nullptr : &fileStack.back()->loc;
                continue;
            }
            return tok;
        }
    }
    void ungetToken(const std::string &s) { ungetValue = s; ungetSet = true; }

    // parser.cpp:413-520 lookupType + :522-700 AddParam
    void addParam(ParamSet &ps, const std::string &decl, const std::vector<double> &nums,
                  const std::vector<std::string> &strs, bool isString) {
        std::istringstream is(decl);
        std::string type, name;
        is >> type >> name;
        if (type.empty() || name.empty()) { Error("Parameter \"%s\" doesn't have a type declaration?!", decl.c_str()); return; }
        bool wantString = (type == "string" || type == "texture" || type == "bool");
        if (wantString && !isString) {
            Error("Expected string parameter value for parameter \"%s\" with type \"%s\". Ignoring.", name.c_str(), type.c_str());
            return;
        }
        if (!wantString && isString && type != "spectrum") {
            Error("Expected numeric parameter value for parameter \"%s\" with type \"%s\".  Ignoring.", name.c_str(), type.c_str());
            return;
        }
        auto toFloats = [&](std::vector<Float> &dst, int mult) {
            size_t n = nums.size();
            if (mult > 1 && n % mult) {
                Warning("Excess values given with %s parameter \"%s\". Ignoring last %d of them.", type.c_str(), name.c_str(), int(n % mult));
                n -= n % mult;
            }
            dst.resize(n);
            for (size_t i = 0; i < n; ++i) dst[i] = nums[i];
        };
        if (type == "integer") { auto &v = ps.ints[name].v; v.resize(nums.size()); for (size_t i = 0; i < nums.size(); ++i) v[i] = int(nums[i]); }
        else if (type == "float") toFloats(ps.floats[name].v, 1);
        else if (type == "bool") {
            auto &v = ps.bools[name].v;
            for (auto &s : strs) {
                if (s == "true") v.push_back(true);
                else if (s == "false") v.push_back(false);
                else { Warning("Value \"%s\" unknown for Boolean parameter \"%s\".Using \"false\".", s.c_str(), name.c_str()); v.push_back(false); }
            }
        }
        else if (type == "point2" || type == "vector2") toFloats(ps.point2s[name].v, 2);
        else if (type == "point3" || type == "point") toFloats(ps.point3s[name].v, 3);
        else if (type == "vector3" || type == "vector") toFloats(ps.vector3s[name].v, 3);
        else if (type == "normal") toFloats(ps.normals[name].v, 3);
        else if (type == "rgb" || type == "color") toFloats(ps.spectra[name].v, 3);
        else if (type == "string") ps.strings[name].v = strs;
        else if (type == "texture") {
            if (strs.size() == 1) ps.textures[name].v = strs;
            else Error("Only one string allowed for \"texture\" parameter \"%s\"", name.c_str());
        }
        else if (type == "xyz") {
            std::vector<Float> v;
            toFloats(v, 3);
            XYZToRGBValues(v, &ps.spectra[name].v);
        } else if (type == "blackbody") {  // (temperature in K, scale) pairs
            std::vector<Float> v;
            if (nums.size() % 2) Warning("Excess value given with blackbody parameter \"%s\". Ignoring extra one.", decl.c_str());
            for (size_t i = 0; i + 1 < nums.size(); i += 2) { v.push_back(nums[i]); v.push_back(nums[i + 1]); }
            BlackbodyToRGBValues(v, &ps.spectra[name].v);
        } else if (type == "spectrum") {  // .spd file names, or inline (wavelength in nm, value) pairs
            if (isString) SpectrumFilesToRGBValues(strs, &ps.spectra[name].v);
            else {
                std::vector<Float> v;
                if (nums.size() % 2) Warning("Non-even number of values given with sampled spectrum parameter \"%s\". Ignoring extra.", decl.c_str());
                for (size_t i = 0; i + 1 < nums.size(); i += 2) { v.push_back(nums[i]); v.push_back(nums[i + 1]); }
                SampledToRGBValues(v, &ps.spectra[name].v);
            }
        }
        else Error("Unable to decode type for name \"%s\"", decl.c_str());
    }

    ParamSet parseParams() {  // parser.cpp:702-781
        ParamSet ps;
        while (true) {
            std::string decl = nextToken(false);
            if (decl.empty()) return ps;
            if (!isQuoted(decl)) { ungetToken(decl); return ps; }
            std::vector<double> nums; std::vector<std::string> strs; bool isString = false;
            auto addVal = [&](const std::string &val) {
                if (isQuoted(val)) {
                    if (!nums.empty()) { Error("mixed string and numeric parameters"); Fatal(); }
                    isString = true; strs.push_back(val.substr(1, val.size() - 2));
                } else if (val[0] == 't' && val == "true") { isString = true; strs.push_back("true"); }
                else if (val[0] == 'f' && val == "false") { isString = true; strs.push_back("false"); }
                else {
                    if (!strs.empty()) { Error("mixed string and numeric parameters"); Fatal(); }
                    nums.push_back(parseNumber(val));
                }
            };
            std::string val = nextToken(true);
            if (val == "[") {
                // (a long, purely numeric array is converted by all threads at once; see Tokenizer::NumberArray)
                if (!(!ungetSet && !fileStack.empty() && fileStack.back()->NumberArray(&nums)))
                    while (true) { val = nextToken(true); if (val == "]") break; addVal(val); }
            } else addVal(val);
            addParam(ps, dequote(decl), nums, strs, isString);
        }
    }

    void run() {  // parser.cpp:862-1090
        struct LocGuard { ~LocGuard() { parserLoc = nullptr; } } locGuard;  // also when a fatal error unwinds through here
        parserLoc = &fileStack.back()->loc;
        auto withParams = [&](std::function<void(const std::string &, const ParamSet &)> fn) {
            std::string n = dequote(nextToken(true));
            ParamSet params = parseParams();
            fn(n, params);
        };
        auto num = [&]() -> Float { return (Float)parseNumber(nextToken(true)); };
        while (true) {
            std::string tok = nextToken(false);
            if (tok.empty()) break;
            if (tok == "AttributeBegin") pbrtAttributeBegin();
            else if (tok == "AttributeEnd") pbrtAttributeEnd();
            else if (tok == "ActiveTransform") {
                std::string a = nextToken(true);
                if (a == "All") pbrtActiveTransformAll();
                else if (a == "EndTime") pbrtActiveTransformEndTime();
                else if (a == "StartTime") pbrtActiveTransformStartTime();
                else { Error("Unexpected token: %s", a.c_str()); Fatal(); }
            }
            else if (tok == "AreaLightSource") withParams(pbrtAreaLightSource);
            else if (tok == "Accelerator") withParams(pbrtAccelerator);
            else if (tok == "ConcatTransform" || tok == "Transform") {
                if (nextToken(true) != "[") { Error("Unexpected token"); Fatal(); }
                Float m[16];
                for (int i = 0; i < 16; ++i) m[i] = num();
                if (nextToken(true) != "]") { Error("Unexpected token"); Fatal(); }
                if (tok == "Transform") pbrtTransform(m); else pbrtConcatTransform(m);
            }
            else if (tok == "CoordinateSystem") pbrtCoordinateSystem(dequote(nextToken(true)));
            else if (tok == "CoordSysTransform") pbrtCoordSysTransform(dequote(nextToken(true)));
            else if (tok == "Camera") withParams(pbrtCamera);
            else if (tok == "Film") withParams(pbrtFilm);
            else if (tok == "Integrator") withParams(pbrtIntegrator);
            else if (tok == "Include") {
                std::string fn = dequote(nextToken(true));
                fn = AbsolutePath(ResolveFilename(fn));
                // a file that includes itself (directly or not) nests without end; the reference reads files until memory runs out
                if (fileStack.size() >= 256) { Error("Include \"%s\": more than 256 nested files (a file that includes itself?)", fn.c_str()); Fatal(); }
                auto t = Tokenizer::FromFile(fn);
                if (t) { fileStack.push_back(std::move(t)); parserLoc = &fileStack.back()->loc; }
            }
            else if (tok == "Identity") pbrtIdentity();
            else if (tok == "LightSource") withParams(pbrtLightSource);
            else if (tok == "LookAt") { Float v[9]; for (int i = 0; i < 9; ++i) v[i] = num(); pbrtLookAt(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]); }
            else if (tok == "MakeNamedMaterial") withParams(pbrtMakeNamedMaterial);
            else if (tok == "MakeNamedMedium") withParams(pbrtMakeNamedMedium);
            else if (tok == "Material") withParams(pbrtMaterial);
            else if (tok == "MediumInterface") {
                std::string a = dequote(nextToken(true));
                std::string b = nextToken(false);
                if (!b.empty() && isQuoted(b)) pbrtMediumInterface(a, dequote(b));
                else { if (!b.empty()) ungetToken(b); pbrtMediumInterface(a, a); }
            }
            else if (tok == "NamedMaterial") pbrtNamedMaterial(dequote(nextToken(true)));
            else if (tok == "ObjectBegin") pbrtObjectBegin(dequote(nextToken(true)));
            else if (tok == "ObjectEnd") pbrtObjectEnd();
            else if (tok == "ObjectInstance") pbrtObjectInstance(dequote(nextToken(true)));
            else if (tok == "PixelFilter") withParams(pbrtPixelFilter);
            else if (tok == "ReverseOrientation") pbrtReverseOrientation();
            else if (tok == "Rotate") { Float v[4]; for (int i = 0; i < 4; ++i) v[i] = num(); pbrtRotate(v[0], v[1], v[2], v[3]); }
            else if (tok == "Shape") {
                const bool timing = getenv("PBRT_HOST_TIMING") != nullptr;
                const auto t0 = std::chrono::steady_clock::now();
                std::string n = dequote(nextToken(true));
                ParamSet params = parseParams();
                const auto t1 = std::chrono::steady_clock::now();
                pbrtShape(n, params);
                if (timing) {
                    const double a = std::chrono::duration<double>(t1 - t0).count(), b = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                    if (a + b > 0.05) fprintf(stderr, "pbrt host: Shape \"%s\": parameters %.3f s, creation %.3f s\n", n.c_str(), a, b);
                }
            }
            else if (tok == "Sampler") withParams(pbrtSampler);
            else if (tok == "Scale") { Float v[3]; for (int i = 0; i < 3; ++i) v[i] = num(); pbrtScale(v[0], v[1], v[2]); }
            else if (tok == "TransformBegin") pbrtTransformBegin();
            else if (tok == "TransformEnd") pbrtTransformEnd();
            else if (tok == "Translate") { Float v[3]; for (int i = 0; i < 3; ++i) v[i] = num(); pbrtTranslate(v[0], v[1], v[2]); }
            else if (tok == "TransformTimes") { Float a = num(), b = num(); pbrtTransformTimes(a, b); }
            else if (tok == "Texture") {
                std::string n = dequote(nextToken(true));
                std::string type = dequote(nextToken(true));
                std::string tex = dequote(nextToken(true));
                ParamSet params = parseParams();
                pbrtTexture(n, type, tex, params);
            }
            else if (tok == "WorldBegin") pbrtWorldBegin();
            else if (tok == "WorldEnd") pbrtWorldEnd();
            else { Error("Unexpected token: %s", tok.c_str()); Fatal(); }
        }
        parserLoc = nullptr;
    }
};
}  // namespace

void pbrtParseFile(const std::string &filename) {  // parser.cpp:1094-1106
    if (filename != "-") SetSearchDirectory(DirectoryContaining(filename));
    auto t = Tokenizer::FromFile(filename);
    if (!t) return;
    Parser p;
    p.fileStack.push_back(std::move(t));
    p.run();
}
void pbrtParseString(const std::string &str) {  // parser.cpp:1108-1112
    Parser p;
    p.fileStack.push_back(Tokenizer::FromString(str));
    p.run();
}
}  // namespace pbrt
