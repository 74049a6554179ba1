// orbslam3_hip/GlueGuard.h — failure policy of the reference-signature glue (integration/*.cc).
//
// The functions those files replace (ORBmatcher::SearchBy*, Optimizer::LocalBundleAdjustment, Frame::ComputeStereoMatches, ...) never
// throw in the reference, and Tracking / LocalMapping / LoopClosing run them on threads that catch nothing: an exception that left one of
// them would terminate the process.  The adapters (include/orbslam3_hip/*.h) do throw — a failing HIP call, a device allocation that does
// not fit, ORB_E_CAPACITY — because a flattened caller wants to know.  Every glue function is therefore a function-try-block whose
// handler ends here: the failure is reported on stderr and counted, the function returns what the reference returns when it found nothing
// (0 matches / 0 inliers / void).  Every glue function does all of its device work BEFORE its first write to the map, so map geometry,
// observations and outlier flags are what they were before the call; the OUTPUT containers are in the function's nothing-found state, which
// for SearchForInitialization / SearchByBoW means vnMatches12 / vpMapPointMatches / vpMatches12 reset (the reference resets them first, too).  A host that wants to react (drop to its CPU bodies, stop the
// session) polls glue_failures().  What "untouched" covers: map geometry, observations, match vectors, outlier flags.  The window-selection
// stamps the reference itself writes while it gathers (mnBALocalForKF / mnBAFixedForKF, IMU::Preintegrated::SetNewBias) are written as
// there; the three Frame-constructor steps (integration/Frame_hip.cc) leave their outputs sized and filled with "nothing found" (-1), which
// is what the constructor code after them indexes.
#ifndef ORBSLAM3_HIP_GLUEGUARD_H
#define ORBSLAM3_HIP_GLUEGUARD_H
#include <atomic>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>

namespace orbslam3_hip {

inline std::atomic<unsigned long>& glue_failure_counter() noexcept {
    static std::atomic<unsigned long> n{0};
    return n;
}
// number of glue calls that ended in the handler since process start.  A host MUST look at it: after a failure LocalBundleAdjustment /
// LocalInertialBA return without having optimised anything and PoseOptimization returns 0 inliers (which Tracking reads as "lost") — under a
// persistent fault (device lost, out of memory) every later call does the same, so poll it after LocalBundleAdjustment / once per frame and
// drop to the CPU bodies or stop the session (integration/README.md), or install a callback with glue_set_failure_callback().
inline unsigned long glue_failures() noexcept { return glue_failure_counter().load(); }

// the most recent failure: the glue function's name and the exception text (what() of the adapters' errors carries the ORB_E_* code), for a
// host that polls.  Tracking, LocalMapping and LoopClosing can fail (and a host thread can poll) at the same time: the slot is written and copied
// out under a mutex — a reader sees one failure's (function, what) pair whole, never a mix of two.
struct GlueLastError { char function[64]; char what[192]; };
inline GlueLastError& glue_last_error_slot() noexcept { static GlueLastError e{{0}, {0}}; return e; }
inline std::mutex& glue_last_error_mutex() noexcept { static std::mutex m; return m; }
inline GlueLastError glue_last_error() noexcept {
    std::lock_guard<std::mutex> lock(glue_last_error_mutex());
    return glue_last_error_slot();
}

// optional notification, called from the failing thread after the counter is bumped: (function name, message, total failures so far).  It runs
// inside a noexcept function on a thread the reference starts bare: it should not throw; if it does, the exception is swallowed here (a report on
// stderr), it never reaches std::terminate
using GlueFailureCallback = void (*)(const char* fn, const char* what, unsigned long total);
inline std::atomic<GlueFailureCallback>& glue_failure_callback_slot() noexcept { static std::atomic<GlueFailureCallback> cb{nullptr}; return cb; }
inline void glue_set_failure_callback(GlueFailureCallback cb) noexcept { glue_failure_callback_slot().store(cb); }

inline int glue_report(const char* fn, const char* what) noexcept {
    unsigned long n;
    {
        std::lock_guard<std::mutex> lock(glue_last_error_mutex());
        GlueLastError& le = glue_last_error_slot();
        std::strncpy(le.function, fn, sizeof(le.function) - 1); le.function[sizeof(le.function) - 1] = 0;
        std::strncpy(le.what, what, sizeof(le.what) - 1); le.what[sizeof(le.what) - 1] = 0;
        n = ++glue_failure_counter();
    }
    // stderr is rate-limited: the first 8 failures, then every 256th — a persistent fault fails at frame rate
    if (n <= 8 || (n & 255u) == 0)
        std::fprintf(stderr, "[orbhip] %s: %s -- call dropped: the reference's nothing-found value is returned (failure %lu%s)\n", fn, what, n,
                     n == 8 ? "; further reports every 256th" : "");
    if (GlueFailureCallback cb = glue_failure_callback_slot().load()) {
        try { cb(fn, what, n); }
        catch (...) { std::fprintf(stderr, "[orbhip] %s: the failure callback threw; ignored\n", fn); }
    }
    return 0;
}
inline int glue_failed(const char* fn, const std::exception& e) noexcept { return glue_report(fn, e.what()); }
inline int glue_failed(const char* fn) noexcept { return glue_report(fn, "unknown exception"); }

}  // namespace orbslam3_hip

// handler of a glue function's function-try-block; `ret` is the statement that leaves the function (`return 0;` / `return;`)
#define ORBHIP_GLUE_CATCH(fn, ret)                                                     \
    catch (const std::exception& e) { (void)orbslam3_hip::glue_failed(fn, e); ret }    \
    catch (...) { (void)orbslam3_hip::glue_failed(fn); ret }
#endif
