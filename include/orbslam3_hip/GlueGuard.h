// orbslam3_hip/GlueGuard.h — failure policy of the reference-signature glue (integration/*.cc).
//
// The functions those files replace (ORBmatcher::SearchBy*, Optimizer::LocalBundleAdjustment, Frame::ComputeStereoMatches, ...) never
// throw in the reference, and Tracking / LocalMapping / LoopClosing run them on threads that catch nothing: an exception that left one of
// them would terminate the process.  The adapters (include/orbslam3_hip/*.h) do throw — a failing HIP call, a device allocation that does
// not fit, ORB_E_CAPACITY — because a flattened caller wants to know.  Every glue function is therefore a function-try-block whose
// handler ends here: the failure is reported on stderr and counted, the function returns what the reference returns when it found nothing
// (0 matches / 0 inliers / void), and — because every glue function does all of its device work BEFORE its first write to the map or the
// frame — the caller's state is exactly what it was before the call.  A host that wants to react (drop to its CPU bodies, stop the
// session) polls glue_failures().  What "untouched" covers: map geometry, observations, match vectors, outlier flags.  The window-selection
// stamps the reference itself writes while it gathers (mnBALocalForKF / mnBAFixedForKF, IMU::Preintegrated::SetNewBias) are written as
// there; the three Frame-constructor steps (integration/Frame_hip.cc) leave their outputs sized and filled with "nothing found" (-1), which
// is what the constructor code after them indexes.
#ifndef ORBSLAM3_HIP_GLUEGUARD_H
#define ORBSLAM3_HIP_GLUEGUARD_H
#include <atomic>
#include <cstdio>
#include <exception>

namespace orbslam3_hip {

inline std::atomic<unsigned long>& glue_failure_counter() noexcept {
    static std::atomic<unsigned long> n{0};
    return n;
}
// number of glue calls that ended in the handler since process start
inline unsigned long glue_failures() noexcept { return glue_failure_counter().load(); }

inline int glue_failed(const char* fn, const std::exception& e) noexcept {
    glue_failure_counter()++;
    std::fprintf(stderr, "[orbhip] %s: %s -- call dropped, caller state untouched\n", fn, e.what());
    return 0;
}
inline int glue_failed(const char* fn) noexcept {
    glue_failure_counter()++;
    std::fprintf(stderr, "[orbhip] %s: unknown exception -- call dropped, caller state untouched\n", fn);
    return 0;
}

}  // namespace orbslam3_hip

// handler of a glue function's function-try-block; `ret` is the statement that leaves the function (`return 0;` / `return;`)
#define ORBHIP_GLUE_CATCH(fn, ret)                                                     \
    catch (const std::exception& e) { (void)orbslam3_hip::glue_failed(fn, e); ret }    \
    catch (...) { (void)orbslam3_hip::glue_failed(fn); ret }
#endif
