// thread_limit.cpp — test-only pthread_create that starts failing with EAGAIN (what a box's thread or
// process limit looks like to std::thread, which then throws std::system_error) after
// RT_TEST_THREAD_BUDGET successful calls.  Linked into bvh_check by tests/test_bvh_build_host.py
// (a root process is exempt from RLIMIT_NPROC, so the limit cannot be imposed from outside).
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>

#include <atomic>
#include <cstdlib>

extern "C" int pthread_create(pthread_t* thread, const pthread_attr_t* attr, void* (*fn)(void*), void* arg) {
    using Fn = int (*)(pthread_t*, const pthread_attr_t*, void* (*)(void*), void*);
    static Fn real = reinterpret_cast<Fn>(dlsym(RTLD_NEXT, "pthread_create"));
    static std::atomic<long> budget{[] {
        const char* e = std::getenv("RT_TEST_THREAD_BUDGET");
        return e ? std::atol(e) : (1L << 40);
    }()};
    if (budget.fetch_sub(1) <= 0) return EAGAIN;
    return real(thread, attr, fn, arg);
}
