// Expression programs compiled at run time (gfx950, hiprtc).
//
// The reference lambdifies the SymPy expressions of an MMSCaseSymbolic for NumPy
// (src/prob1base.py:1283-1347).  tedd_prog.hpp interprets the lowered expressions word by word with the
// registers of a thread in LDS; here the same validated words are spelled out as straight-line C++ -- one
// statement per word, registers as local doubles -- and compiled for the device the context runs on, so that the
// registers live in VGPRs and the compiler schedules the whole program.  The generated translation unit includes
// tedd_prog.hpp from the source directory that ships beside the library and instantiates the very functions the
// interpreter's functors call (prog_forcing_node, prog_exact_node) with its own evaluator: every operation is a
// statement of its own (no contraction across words), the elementary functions are the device library's, so the
// arrays it produces are bit-identical to the interpreter's.
//
// Code objects are cached per process by their source text: the contexts of a convergence sweep (one per mesh
// level, same expressions) compile once.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "tedd_prog.hpp"

namespace tedd {

inline std::string gen_literal(double v) {
    if (v != v) return "__builtin_nan(\"\")";
    if (std::isinf(v)) return v > 0 ? "__builtin_huge_val()" : "(-__builtin_huge_val())";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%a", v);  // hexadecimal floating literal: exact
    return buf;
}

// body of one program: `double r0 = t, r1 = x, r2 = y` are declared by the caller's frame
inline void gen_program_body(std::string& o, const HostProg& P) {
    auto R = [](int k) { return "r" + std::to_string(k); };
    o += "        double";
    for (int k = 3; k < P.n_regs; ++k) o += (k == 3 ? " " : ", ") + R(k) + " = 0.0";
    if (P.n_regs <= 3) o += " unused_ = 0.0";
    o += ";\n";
    for (uint32_t w : P.code) {
        const int op = (int)(w & 0xffu), d = (int)((w >> 8) & 0xffu), a = (int)((w >> 16) & 0xffu), b = (int)(w >> 24);
        const std::string D = R(d), A = R(a), B = R(b);
        o += "        ";
        switch (op) {
            case POP_CONST: o += D + " = " + gen_literal(P.consts[(size_t)(a | (b << 8))]) + ";"; break;
            case POP_OUT: o += "out[" + std::to_string(d) + "] = " + A + ";"; break;
            case POP_ADD: o += D + " = " + A + " + " + B + ";"; break;
            case POP_SUB: o += D + " = " + A + " - " + B + ";"; break;
            case POP_MUL: o += D + " = " + A + " * " + B + ";"; break;
            case POP_DIV: o += D + " = " + A + " / " + B + ";"; break;
            case POP_NEG: o += D + " = -" + A + ";"; break;
            case POP_POWI: {
                // the interpreter's square-and-multiply, unrolled for this exponent (same products in the same order)
                int n = (int)(int8_t)b;
                const bool inv = n < 0;
                n = inv ? -n : n;
                o += "{ double base_ = " + A + "; double p_ = 1.0;";
                while (n) {
                    if (n & 1) o += " p_ = p_ * base_;";
                    n >>= 1;
                    if (n) o += " base_ = base_ * base_;";
                }
                if (inv) o += " p_ = 1.0 / p_;";
                o += " " + D + " = p_; }";
                break;
            }
            case POP_POW: o += D + " = pow(" + A + ", " + B + ");"; break;
            case POP_EXP: o += D + " = exp(" + A + ");"; break;
            case POP_SIN: o += D + " = sin(" + A + ");"; break;
            case POP_COS: o += D + " = cos(" + A + ");"; break;
            case POP_SQRT: o += D + " = sqrt(" + A + ");"; break;
            case POP_LOG: o += D + " = log(" + A + ");"; break;
            case POP_ABS: o += D + " = fabs(" + A + ");"; break;
            case POP_SIGN: o += D + " = (" + A + " > 0.0) ? 1.0 : (" + A + " < 0.0) ? -1.0 : " + A + ";"; break;
            case POP_TANH: o += D + " = tanh(" + A + ");"; break;
            default: o += D + " = " + A + ";"; break;  // POP_MOV
        }
        o += "\n";
    }
}

// the translation unit: evaluator + the two kernels (64 x 4 threads, one node each)
inline std::string gen_source(const HostProg& node, const HostProg& cell, int react) {
    std::string s;
    s += "#include \"tedd_prog.hpp\"\n";
    s += "using namespace tedd;\n";
    // the headers found at run time must be those this library was built from: the kernels receive GridDev, ModelC,
    // ProgAxes and Out5 by value from code compiled here
    s += "static_assert(kProgContractVersion == " + std::to_string(kProgContractVersion) +
         " && sizeof(GridDev) == " + std::to_string(sizeof(GridDev)) + " && sizeof(ModelC) == " + std::to_string(sizeof(ModelC)) +
         " && sizeof(ProgAxes) == " + std::to_string(sizeof(ProgAxes)) + " && sizeof(Out5) == " + std::to_string(sizeof(Out5)) +
         ", \"the headers beside the library are not the ones it was built from\");\n";
    s += "struct GenEval {\n";
    s += "    __device__ __forceinline__ void node(double t, double x, double y, double* out) const {\n";
    s += "        double r0 = t, r1 = x, r2 = y;\n        (void)r0; (void)r1; (void)r2;\n";
    gen_program_body(s, node);
    s += "    }\n";
    s += "    __device__ __forceinline__ void cell(double t, double x, double y, double* out) const {\n";
    s += "        double r0 = t, r1 = x, r2 = y;\n        (void)r0; (void)r1; (void)r2;\n";
    gen_program_body(s, cell);
    s += "    }\n};\n";
    s += "extern \"C\" __global__ void __launch_bounds__(256) tedd_gen_forcing(const GridDev g, const ModelC m, const ProgAxes ax, "
         "double t, const Out5 out, int r0, int r1) {\n";
    s += "    const int j = blockIdx.x * 64 + threadIdx.x, li = r0 + blockIdx.y * 4 + threadIdx.y;\n";
    s += "    if (j < g.ncols && li < r1) prog_forcing_node<" + std::to_string(react) + ">(GenEval{}, g, m, ax, t, out, li, j);\n}\n";
    s += "extern \"C\" __global__ void __launch_bounds__(256) tedd_gen_exact(const GridDev g, const ProgAxes ax, double t, "
         "const Out5 out, int what, int r0, int r1) {\n";
    s += "    const int j = blockIdx.x * 64 + threadIdx.x, li = r0 + blockIdx.y * 4 + threadIdx.y;\n";
    s += "    if (j < g.ncols && li < r1) prog_exact_node(GenEval{}, g, ax, t, out, what, li, j);\n}\n";
    return s;
}

// directory of the headers the generated source includes: TEDD_CSRC_DIR, else ../csrc beside this library
inline std::string gen_include_dir() {
    auto has_header = [](const std::string& dir) {
        FILE* f = std::fopen((dir + "/tedd_prog.hpp").c_str(), "r");
        if (f) std::fclose(f);
        return f != nullptr;
    };
    if (const char* e = std::getenv("TEDD_CSRC_DIR")) {
        if (has_header(e)) return e;
    }
    Dl_info info;
    if (dladdr((const void*)&gen_literal, &info) && info.dli_fname) {
        std::string lib = info.dli_fname;
        const size_t cut = lib.find_last_of('/');
        const std::string dir = (cut == std::string::npos ? std::string(".") : lib.substr(0, cut)) + "/../csrc";
        if (has_header(dir)) return dir;
    }
    throw std::runtime_error("generated MMS kernels: the library's source headers (tedd_prog.hpp) were not found beside "
                             "it (../csrc) nor under TEDD_CSRC_DIR");
}

// The run-time compiler is loaded when the first program is compiled, not when the library is: a machine without
// it keeps the interpreter (and everything that is not an expression program).
struct HiprtcApi {
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    decltype(&hiprtcGetErrorString) error_string = nullptr;
    // (call with the compile mutex held)
    static const HiprtcApi& get() {
        static HiprtcApi api;
        static bool tried = false;
        static std::string why;
        if (!tried) {
            tried = true;
            void* h = nullptr;
            for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
            if (!h) {
                why = "generated MMS kernels: the run-time compiler library (libhiprtc.so) could not be loaded";
            } else {
                auto sym = [&](const char* n) {
                    void* p = dlsym(h, n);
                    if (!p && why.empty()) why = std::string("generated MMS kernels: libhiprtc.so lacks ") + n;
                    return p;
                };
                api.create = reinterpret_cast<decltype(api.create)>(sym("hiprtcCreateProgram"));
                api.compile = reinterpret_cast<decltype(api.compile)>(sym("hiprtcCompileProgram"));
                api.log_size = reinterpret_cast<decltype(api.log_size)>(sym("hiprtcGetProgramLogSize"));
                api.log = reinterpret_cast<decltype(api.log)>(sym("hiprtcGetProgramLog"));
                api.code_size = reinterpret_cast<decltype(api.code_size)>(sym("hiprtcGetCodeSize"));
                api.code = reinterpret_cast<decltype(api.code)>(sym("hiprtcGetCode"));
                api.destroy = reinterpret_cast<decltype(api.destroy)>(sym("hiprtcDestroyProgram"));
                api.error_string = reinterpret_cast<decltype(api.error_string)>(sym("hiprtcGetErrorString"));
            }
        }
        if (!why.empty()) throw std::runtime_error(why);
        return api;
    }
};

// compiled code object of a source text for an architecture (cached per process)
inline const std::vector<char>& gen_compile(const std::string& src, const std::string& arch) {
    static std::mutex mu;
    static std::map<std::string, std::vector<char>> cache;
    std::lock_guard<std::mutex> lk(mu);
    // (the header directory is part of the key: the same source against another tree is another program)
    const std::string inc = "-I" + gen_include_dir();
    const std::string key = arch + "\n" + inc + "\n" + src;
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const HiprtcApi& rtc = HiprtcApi::get();
    const std::string march = "--offload-arch=" + arch;
    hiprtcProgram prog = nullptr;
    if (rtc.create(&prog, src.c_str(), "tedd_generated_mms.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        throw std::runtime_error("generated MMS kernels: hiprtcCreateProgram failed");
    // -ffp-contract=on as the library itself: a multiply and an add fuse only inside one source expression
    const char* opts[] = {march.c_str(), "-O3", "-std=c++17", "-ffp-contract=on", inc.c_str()};
    const hiprtcResult rc = rtc.compile(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        std::string log;
        if (rtc.log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) {
            log.resize(n);
            (void)rtc.log(prog, &log[0]);
        }
        (void)rtc.destroy(&prog);
        if (log.size() > 1500) log.resize(1500);
        throw std::runtime_error(std::string("generated MMS kernels: compilation failed: ") + rtc.error_string(rc) + "\n" + log);
    }
    size_t size = 0;
    std::vector<char> code;
    if (rtc.code_size(prog, &size) != HIPRTC_SUCCESS || size == 0) {
        (void)rtc.destroy(&prog);
        throw std::runtime_error("generated MMS kernels: empty code object");
    }
    code.resize(size);
    const hiprtcResult rc2 = rtc.code(prog, code.data());
    (void)rtc.destroy(&prog);
    if (rc2 != HIPRTC_SUCCESS) throw std::runtime_error("generated MMS kernels: hiprtcGetCode failed");
    return cache.emplace(key, std::move(code)).first->second;
}

}  // namespace tedd
