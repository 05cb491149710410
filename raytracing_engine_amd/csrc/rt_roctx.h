// rt_roctx.h — roctx ranges around the stage launches (profiling row: "roctx ranges per stage"; the
// reference's only instrumentation is the FPS print, src/main.rs:719,730).  The marker library is opened lazily with dlopen and
// only when RT_ROCTX is set to something other than "0" (librocprofiler-sdk-roctx, then the older libroctx64), so the default
// path carries no profiler dependency; see the ranges with  RT_ROCTX=1 rocprofv3 --marker-trace --kernel-trace -- <program>.
// A range brackets the host-side enqueue of a stage; the tool ties the kernels launched inside it to the range.
#pragma once
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

namespace rt {

struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    static const RoctxApi& get() {
        static const RoctxApi api = [] {
            RoctxApi a;
            const char* on = std::getenv("RT_ROCTX");
            if (!on || (on[0] == '0' && on[1] == 0)) return a;
            for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
                if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                    a.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                    a.pop = (int (*)())dlsym(h, "roctxRangePop");
                    if (a.push && a.pop) return a;
                    a.push = nullptr;
                    a.pop = nullptr;
                }
            }
            return a;
        }();
        return api;
    }
};

struct RoctxRange {  // RAII: one nested range
    bool open = false;
    explicit RoctxRange(const char* name) {
        const RoctxApi& a = RoctxApi::get();
        if (a.push) {
            a.push(name);
            open = true;
        }
    }
    RoctxRange(const char* name, unsigned index) {
        const RoctxApi& a = RoctxApi::get();
        if (a.push) {
            char buf[64];
            std::snprintf(buf, sizeof buf, "%s %u", name, index);
            a.push(buf);
            open = true;
        }
    }
    ~RoctxRange() {
        if (open) RoctxApi::get().pop();
    }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

inline bool roctx_active() { return RoctxApi::get().push != nullptr; }  // test hook (rt_get_stats is not extended for it)

}  // namespace rt
