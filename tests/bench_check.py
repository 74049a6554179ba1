"""Checker for the benchmark's own configuration — TEST INFRASTRUCTURE (used by tests/test_bench_config_parity.py and, after the timed
region, by bench.py's `parity_checked_frames`; never inside a timed region, never by the product).

compare_step(): the frames of one GPU step (ORBextractor -> UndistortKeyPoints -> AssignFeaturesToGrid -> SearchByProjection, as bench.py's
StepPipeline runs them) against the oracle's per-frame unit on the same images and the same prepared projection records
(oracle/bench_oracle.cpp oro_extract_match_frames_mt; reference ORBextractor.cc:1074-1156, Frame.cc:874-924, ORBmatcher.cc:2244-2509).
Everything is compared bit for bit: {N, monoIndex}, the 28-byte key point records, the descriptors, the undistorted records, the per-query
match, the mvpMapPoints image and the match count."""
import numpy as np

import oracle_lib as O


def compare_step(frames, snap, q, qdesc, nq, cam9, grid4, nfeatures, lap=(0, 1000), mode=1, th_dist=100, nnratio=0.9, check_ori=True,
                 nthreads=None, do_match=True, max_report=8):
    """frames [n,H,W] u8 and every array of `snap` / q / qdesc / nq hold the SAME n frames (already gathered by the caller).
    snap: dict(kps [n,cap,7] f32, desc [n,cap,32] u8, counts [n,2] i32 and — with do_match — un [n,cap,7], q_match [n,cap_q], kp_match [n,cap], nm [n]).
    -> (frames compared, list of mismatch descriptions (empty = bit-identical), dict of totals)"""
    n = len(frames)
    cap = snap["kps"].shape[1]
    o = O.extract_match_frames(frames, np.arange(n, dtype=np.int32), cam9, grid4, q, qdesc, nq, mode, th_dist, nnratio, check_ori, do_match,
                               nfeatures=nfeatures, lap=lap, nthreads=nthreads, cap=cap)
    bad = []

    def rep(i, what, detail=""):
        if len(bad) < max_report:
            bad.append("frame %d: %s %s" % (i, what, detail))
        elif len(bad) == max_report:
            bad.append("...")

    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
    tot = dict(keypoints=0, matches=0)
    for i in range(n):
        no, mo = int(o["counts"][i, 0]), int(o["counts"][i, 1])
        ng, mg = int(snap["counts"][i, 0]), int(snap["counts"][i, 1])
        tot["keypoints"] += no
        if (no, mo) != (ng, mg):
            rep(i, "{N, monoIndex}", "oracle %s, device %s" % ((no, mo), (ng, mg)))
            continue
        if not np.array_equal(bits(snap["kps"][i, :no]), bits(o["kps"][i, :no])):
            rep(i, "key point records", "first differing row %d" % int(np.nonzero((bits(snap["kps"][i, :no]) != bits(o["kps"][i, :no])).any(1))[0][0]))
            continue
        if not np.array_equal(snap["desc"][i, :no], o["desc"][i, :no]):
            rep(i, "descriptors", "%d rows differ" % int((snap["desc"][i, :no] != o["desc"][i, :no]).any(1).sum()))
            continue
        if not do_match:
            continue
        tot["matches"] += int(o["nm"][i])
        if not np.array_equal(bits(snap["un"][i, :no]), bits(o["un"][i, :no])):
            rep(i, "undistorted records")
            continue
        if int(snap["nm"][i]) != int(o["nm"][i]):
            rep(i, "nmatches", "oracle %d, device %d" % (int(o["nm"][i]), int(snap["nm"][i])))
            continue
        if not np.array_equal(snap["kp_match"][i, :no], o["kp_match"][i, :no]):
            rep(i, "mvpMapPoints", "%d entries differ" % int((snap["kp_match"][i, :no] != o["kp_match"][i, :no]).sum()))
            continue
        k = int(nq[i])
        if not np.array_equal(snap["q_match"][i, :k], o["q_match"][i, :k]):
            rep(i, "per-query match", "%d entries differ" % int((snap["q_match"][i, :k] != o["q_match"][i, :k]).sum()))
    return n, bad, tot
