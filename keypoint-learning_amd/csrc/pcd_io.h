// pcd_io.h -- PCD reader (ascii, binary, binary_compressed; float32 x y z [normal_x normal_y normal_z]) and
// ascii writer for the command-line programs of this package: what pcl::io::loadPCDFile /
// savePCDFileASCII do for /root/reference/src/main_test_detector.cpp:143 and :212-216.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "KeypointLearning.h"

namespace kpl_io {

typedef pcl::PointXYZ PointInT;
typedef pcl::Normal PointNormalT;
typedef pcl::PointXYZI KeypointT;

// LZF (the codec of PCD "binary_compressed"): a control byte < 32 starts a run of ctrl + 1 literals;
// otherwise it is a back reference of length (ctrl >> 5) + 2 (7 = extended by the next byte) at
// distance ((ctrl & 31) << 8 | next byte) + 1
inline bool lzf_decompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            const size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_len) return false;
            memcpy(out + op, in + ip, run);
            ip += run;
            op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) {
                if (ip >= in_len) return false;
                len += in[ip++];
            }
            if (ip >= in_len) return false;
            const size_t dist = ((size_t)(ctrl & 31) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_len) return false;
            for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];     // may overlap itself
        }
    }
    return op == out_len;
}

inline bool load_pcd(const std::string &path, pcl::PointCloud<PointInT> &cloud, pcl::PointCloud<PointNormalT> &normals) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); return false; }
    std::vector<std::string> fields;
    std::vector<int> sizes, counts;
    std::vector<char> types;
    size_t npoints = 0, width = 0, height = 0;
    float viewpoint[3] = {0.f, 0.f, 0.f};
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "FIELDS") { std::string s; while (ls >> s) fields.push_back(s); }
        else if (key == "SIZE") { int s; while (ls >> s) sizes.push_back(s); }
        else if (key == "TYPE") { char c; while (ls >> c) types.push_back(c); }
        else if (key == "COUNT") { int c; while (ls >> c) counts.push_back(c); }
        else if (key == "WIDTH") ls >> width;
        else if (key == "HEIGHT") ls >> height;
        else if (key == "POINTS") ls >> npoints;
        else if (key == "VIEWPOINT") {                // tx ty tz qw qx qy qz: the translation is the sensor origin
            float t[3] = {0.f, 0.f, 0.f};
            if (ls >> t[0] >> t[1] >> t[2]) { viewpoint[0] = t[0]; viewpoint[1] = t[1]; viewpoint[2] = t[2]; }
        }
        else if (key == "DATA") { ls >> data_kind; break; }
    }
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size()) { fprintf(stderr, "%s: bad PCD header\n", path.c_str()); return false; }
    auto find = [&](const char *n) { for (size_t i = 0; i < fields.size(); ++i) if (fields[i] == n) return (int)i; return -1; };
    const int ix = find("x"), iy = find("y"), iz = find("z");
    const int inx = find("normal_x"), iny = find("normal_y"), inz = find("normal_z");
    if (ix < 0 || iy < 0 || iz < 0) { fprintf(stderr, "%s: no x y z fields\n", path.c_str()); return false; }
    std::vector<size_t> offset(fields.size());
    size_t rec = 0, ncols = 0;
    std::vector<size_t> col(fields.size());
    for (size_t i = 0; i < fields.size(); ++i) { offset[i] = rec; col[i] = ncols; rec += (size_t)sizes[i] * counts[i]; ncols += counts[i]; }
    for (int i : {ix, iy, iz, inx, iny, inz}) if (i >= 0 && (sizes[i] != 4 || types[i] != 'F')) { fprintf(stderr, "%s: only float32 coordinates are supported\n", path.c_str()); return false; }
    cloud.clear();
    normals.clear();
    const bool has_n = inx >= 0 && iny >= 0 && inz >= 0;
    if (data_kind == "ascii") {
        std::vector<double> row(ncols);
        for (size_t p = 0; p < npoints; ++p) {
            for (size_t c = 0; c < ncols; ++c) {
                std::string tok;
                if (!(f >> tok)) { fprintf(stderr, "%s: truncated data\n", path.c_str()); return false; }
                row[c] = (tok == "nan" || tok == "NaN") ? NAN : atof(tok.c_str());
            }
            cloud.push_back(PointInT((float)row[col[ix]], (float)row[col[iy]], (float)row[col[iz]]));
            if (has_n) { PointNormalT n; n.normal_x = (float)row[col[inx]]; n.normal_y = (float)row[col[iny]]; n.normal_z = (float)row[col[inz]]; normals.push_back(n); }
        }
    } else if (data_kind == "binary") {
        std::vector<char> buf(rec * npoints);
        f.read(buf.data(), (std::streamsize)buf.size());
        if ((size_t)f.gcount() != buf.size()) { fprintf(stderr, "%s: truncated data\n", path.c_str()); return false; }
        auto get = [&](size_t p, int fi) { float v; memcpy(&v, &buf[p * rec + offset[fi]], 4); return v; };
        for (size_t p = 0; p < npoints; ++p) {
            cloud.push_back(PointInT(get(p, ix), get(p, iy), get(p, iz)));
            if (has_n) { PointNormalT n; n.normal_x = get(p, inx); n.normal_y = get(p, iny); n.normal_z = get(p, inz); normals.push_back(n); }
        }
    } else if (data_kind == "binary_compressed") {
        // uint32 compressed size, uint32 uncompressed size, LZF stream; the payload is field-major
        // (all x, then all y, ...) instead of point-major
        uint32_t csize = 0, usize = 0;
        f.read((char *)&csize, 4);
        f.read((char *)&usize, 4);
        if (!f || usize != rec * npoints) { fprintf(stderr, "%s: bad binary_compressed sizes\n", path.c_str()); return false; }
        std::vector<unsigned char> in(csize), buf(usize);
        f.read((char *)in.data(), (std::streamsize)in.size());
        if ((size_t)f.gcount() != in.size() || !lzf_decompress(in.data(), in.size(), buf.data(), buf.size())) {
            fprintf(stderr, "%s: corrupt binary_compressed data\n", path.c_str());
            return false;
        }
        std::vector<size_t> plane(fields.size());     // start of each field's block
        size_t at = 0;
        for (size_t i = 0; i < fields.size(); ++i) { plane[i] = at; at += (size_t)sizes[i] * counts[i] * npoints; }
        auto get = [&](size_t p, int fi) { float v; memcpy(&v, &buf[plane[fi] + p * 4 * counts[fi]], 4); return v; };
        for (size_t p = 0; p < npoints; ++p) {
            cloud.push_back(PointInT(get(p, ix), get(p, iy), get(p, iz)));
            if (has_n) { PointNormalT n; n.normal_x = get(p, inx); n.normal_y = get(p, iny); n.normal_z = get(p, inz); normals.push_back(n); }
        }
    } else {
        fprintf(stderr, "%s: PCD DATA '%s' is not supported (ascii, binary and binary_compressed are)\n", path.c_str(), data_kind.c_str());
        return false;
    }
    cloud.is_dense = true;
    for (auto &p : cloud.points) if (!pcl::isFinite(p)) cloud.is_dense = false;
    for (int k = 0; k < 3; ++k) cloud.sensor_origin_[k] = viewpoint[k];    // pcl::PCDReader: sensor_origin_ = VIEWPOINT translation
    if (height > 1 && width * height == npoints) {      // an organized cloud keeps its image shape, as with pcl::PCDReader
        cloud.width = (uint32_t)width;
        cloud.height = (uint32_t)height;
    }
    return true;
}

inline bool save_pcd_ascii(const std::string &path, const pcl::PointCloud<KeypointT> &kp) {   // :212-216
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return false;
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
               "COUNT 1 1 1 1\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA ascii\n", kp.size(), kp.size());
    for (auto &p : kp.points) fprintf(f, "%.9g %.9g %.9g %.9g\n", p.x, p.y, p.z, p.intensity);
    fclose(f);
    return true;
}

}  // namespace kpl_io
