// tools/opencv_crosscheck.cpp — the C++ half of tools/opencv_crosscheck.py: pins the oracle's OpenCV-delegated steps against the OpenCV the reference
// is actually built with, and (optionally) against the reference's own ORBextractor.  CANNOT be built in this repository's image (no OpenCV): it is
// for a maintainer's machine.  Nothing in the product, the tests or bench.py uses it.
//
//   python tools/opencv_crosscheck.py dump DIR                                   (in this repository: the oracle's per-stage outputs)
//   g++ -std=c++11 -O2 tools/opencv_crosscheck.cpp `pkg-config --cflags --libs opencv` -o opencv_crosscheck      (stages only), or with the reference:
//   g++ -std=c++11 -O2 -DWITH_REFERENCE_EXTRACTOR -I<ORB_SLAM3>/include tools/opencv_crosscheck.cpp <ORB_SLAM3>/src/ORBextractor.cc
//       `pkg-config --cflags --libs opencv` -o opencv_crosscheck
//   ./opencv_crosscheck DIR
//
// Per case and pyramid level every stage is recomputed with real OpenCV FROM THE ORACLE'S INPUT of that stage (a divergence belongs to one stage):
// cv::resize INTER_LINEAR (reference src/ORBextractor.cc:1171), copyMakeBorder BORDER_REFLECT_101 (:1173-1179), per-cell cv::FAST with the minThFAST
// retry (:763-855), GaussianBlur 7x7 sigma 2 (:1121), IC_Angle with cv::fastAtan2 (:75-102); with the reference compiled in, ORBextractor::operator()
// (:1074-1156) on image.pgm against keypoints.bin / descriptors.bin (28-byte cv::KeyPoint records compared as bytes).  Prints the first divergence
// per stage; exit code = number of divergent stages.  Expected to agree: OpenCV 3.2 / 3.3.  OpenCV >= 3.4.1 blurs CV_8U in fixed point
// (bit-exact smooth): expect the GaussianBlur stage — and with it a few descriptor bits of the reference itself — to differ there.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#ifdef WITH_REFERENCE_EXTRACTOR
#include "ORBextractor.h"
#endif

static cv::Mat readPgm(const std::string& path) {
    std::ifstream f(path.c_str(), std::ios::binary);
    std::string magic;
    int w = 0, h = 0, mx = 0;
    f >> magic >> w >> h >> mx;
    f.get();
    cv::Mat m;
    if (!f || magic != "P5" || mx != 255) return m;
    m.create(h, w, CV_8UC1);
    f.read((char*)m.data, (std::streamsize)w * h);
    return f ? m : cv::Mat();
}

static std::string firstDiff(const cv::Mat& a, const cv::Mat& b) {   // a: OpenCV, b: oracle
    if (a.size() != b.size()) return "sizes differ";
    long n = 0;
    int fx = -1, fy = -1;
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < a.cols; x++)
            if (a.at<uchar>(y, x) != b.at<uchar>(y, x)) { if (!n) { fx = x; fy = y; } n++; }
    if (!n) return "";
    std::ostringstream s;
    s << n << " of " << (long)a.rows * a.cols << " pixels differ; first at (x=" << fx << ", y=" << fy << "): OpenCV " << (int)a.at<uchar>(fy, fx) << ", oracle "
      << (int)b.at<uchar>(fy, fx);
    return s.str();
}

static int report(const char* stage, int level, const std::string& msg) {
    std::printf("  %-34s level %d: %s\n", stage, level, msg.empty() ? "identical" : msg.c_str());
    return msg.empty() ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc != 2) { std::printf("usage: %s DIR   (written by tools/opencv_crosscheck.py dump DIR)\n", argv[0]); return 255; }
    std::printf("OpenCV %s\n", CV_VERSION);
    const char* cases[] = {"extract_320x240", "extract_400x300_lap", "euroc_752x480"};
    int bad = 0;
    for (const char* name : cases) {
        const std::string d = std::string(argv[1]) + "/" + name + "/";
        std::ifstream cf((d + "config.txt").c_str());
        int W, H, nf, nl, ini, mn, lap0, lap1, mono, nk, umax[16];
        double scale;
        cf >> W >> H >> nf >> scale >> nl >> ini >> mn >> lap0 >> lap1 >> mono >> nk;
        for (int i = 0; i < 16; i++) cf >> umax[i];
        if (!cf) { std::printf("%s: cannot read config.txt\n", name); bad++; continue; }
        std::printf("%s\n", name);
        std::vector<cv::Mat> lev(nl);
        for (int l = 0; l < nl; l++) lev[l] = readPgm(d + "level_" + std::to_string(l) + ".pgm");
        for (int l = 0; l < nl; l++) {
            const std::string L = std::to_string(l);
            if (l) {
                cv::Mat r;
                cv::resize(lev[l - 1], r, lev[l].size(), 0, 0, cv::INTER_LINEAR);
                bad += report("cv::resize INTER_LINEAR", l, firstDiff(r, lev[l]));
            }
            cv::Mat B;
            cv::copyMakeBorder(lev[l], B, 19, 19, 19, 19, cv::BORDER_REFLECT_101);
            bad += report("cv::copyMakeBorder REFLECT_101", l, firstDiff(B, readPgm(d + "bordered_" + L + ".pgm")));
            {   // the cell loop of ComputeKeyPointsOctTree, ORBextractor.cc:763-855
                std::set<std::tuple<int, int, int> > got, want;
                std::ifstream c((d + "candidates_" + L + ".txt").c_str());
                int x, y, s;
                while (c >> x >> y >> s) want.insert(std::make_tuple(x, y, s));
                const int minBX = 16, minBY = 16, maxBX = lev[l].cols - 16, maxBY = lev[l].rows - 16;
                const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
                const int nCols = (int)(width / 30), nRows = (int)(height / 30);
                const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
                for (int i = 0; i < nRows; i++) {
                    const int iniY = minBY + i * hCell;
                    int maxY = iniY + hCell + 6;
                    if (iniY >= maxBY - 3) continue;
                    if (maxY > maxBY) maxY = maxBY;
                    for (int j = 0; j < nCols; j++) {
                        const int iniX = minBX + j * wCell;
                        int maxX = iniX + wCell + 6;
                        if (iniX >= maxBX - 6) continue;
                        if (maxX > maxBX) maxX = maxBX;
                        std::vector<cv::KeyPoint> v;
                        cv::FAST(lev[l].rowRange(iniY, maxY).colRange(iniX, maxX), v, ini, true);
                        if (v.empty()) cv::FAST(lev[l].rowRange(iniY, maxY).colRange(iniX, maxX), v, mn, true);
                        for (const cv::KeyPoint& k : v) got.insert(std::make_tuple((int)k.pt.x + j * wCell, (int)k.pt.y + i * hCell, (int)k.response));
                    }
                }
                std::string msg;
                if (got != want) {
                    size_t onlyCv = 0, onlyOr = 0;
                    for (const auto& t : got) onlyCv += !want.count(t);
                    for (const auto& t : want) onlyOr += !got.count(t);
                    msg = std::to_string(onlyCv) + " candidates only in OpenCV, " + std::to_string(onlyOr) + " only in the oracle";
                }
                bad += report("cv::FAST per cell (+ retry)", l, msg);
            }
            const cv::Mat ob = readPgm(d + "blurred_" + L + ".pgm");
            if (!ob.empty()) {
                cv::Mat g = lev[l].clone();
                cv::GaussianBlur(g, g, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
                std::string m = firstDiff(g, ob);
                if (!m.empty()) m += "   [OpenCV >= 3.4.1: fixed-point path]";
                bad += report("cv::GaussianBlur 7x7 sigma 2", l, m);
            }
            {   // IC_Angle, ORBextractor.cc:75-102, on the bordered level
                std::ifstream a((d + "angles_" + L + ".txt").c_str());
                long x, y, bits, n = 0, nb = 0;
                while (a >> x >> y >> bits) {
                    const uchar* center = &B.at<uchar>((int)y + 19, (int)x + 19);
                    const int step = (int)B.step1();
                    int m_01 = 0, m_10 = 0;
                    for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
                    for (int v = 1; v <= 15; ++v) {
                        int v_sum = 0;
                        const int dd = umax[v];
                        for (int u = -dd; u <= dd; ++u) {
                            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
                            v_sum += (val_plus - val_minus);
                            m_10 += u * (val_plus + val_minus);
                        }
                        m_01 += v * v_sum;
                    }
                    const float ang = cv::fastAtan2((float)m_01, (float)m_10);
                    unsigned int ub;
                    std::memcpy(&ub, &ang, 4);
                    nb += ub != (unsigned int)bits;
                    n++;
                }
                bad += report("IC_Angle / cv::fastAtan2", l, nb ? std::to_string(nb) + " of " + std::to_string(n) + " angles differ in their bits" : "");
            }
        }
#ifdef WITH_REFERENCE_EXTRACTOR
        {
            static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
            ORB_SLAM3::ORBextractor ex(nf, (float)scale, nl, ini, mn);
            std::vector<cv::KeyPoint> k;
            cv::Mat desc;
            std::vector<int> lap = {lap0, lap1};
            const int m = ex(readPgm(d + "image.pgm"), cv::Mat(), k, desc, lap);
            std::vector<char> ok((size_t)nk * 28), od((size_t)nk * 32);
            std::ifstream fk((d + "keypoints.bin").c_str(), std::ios::binary), fd((d + "descriptors.bin").c_str(), std::ios::binary);
            fk.read(ok.data(), (std::streamsize)ok.size());
            fd.read(od.data(), (std::streamsize)od.size());
            std::string msg;
            if (m != mono || (int)k.size() != nk) msg = "monoIndex / count: reference " + std::to_string(m) + " / " + std::to_string(k.size()) + ", oracle " + std::to_string(mono) + " / " + std::to_string(nk);
            else {
                long kb = 0, db = 0, first = -1;
                for (int i = 0; i < nk; i++) {
                    const bool dk = std::memcmp(&k[i], ok.data() + (size_t)i * 28, 28) != 0, dd = std::memcmp(desc.ptr(i), od.data() + (size_t)i * 32, 32) != 0;
                    kb += dk; db += dd;
                    if ((dk || dd) && first < 0) first = i;
                }
                if (kb || db) msg = std::to_string(kb) + " key point records and " + std::to_string(db) + " descriptors differ; first at output index " + std::to_string(first) +
                                    "  (octree ties: oracle rule R1, sin / cos: rule R2 — oracle/orb_oracle.cpp header)";
            }
            bad += report("reference ORBextractor::operator()", 0, msg);
        }
#endif
    }
    std::printf("divergent stages: %d\n", bad);
    return bad > 254 ? 254 : bad;
}
