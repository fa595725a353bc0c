// node_sim: drives the node shims (node_shim.hpp) on a raw frame file the way the ROS nodes would be driven, and
// prints what they would publish as one JSON line.  Used by tests/test_gpu_node_sim.py.
//   node_sim pose    <bgr.raw> <w> <h> <roi_x> <roi_y> <roi_w> <roi_h> <cam_roi_x> <cam_roi_y> <Hmin Smin Vmin Hmax Smax Vmax> <mask_out.raw>
//   node_sim polygon <bgr.raw> <w> <h> <Hmin Smin Vmin Hmax Smax Vmax> <mask_out.raw> <proc_out.raw>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "node_shim.hpp"

static std::vector<uint8_t> slurp(const char *path, size_t n)
{
    std::vector<uint8_t> v(n);
    std::ifstream f(path, std::ios::binary);
    if (!f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)n)) throw std::runtime_error(std::string("short read: ") + path);
    return v;
}
static void dump(const char *path, const std::vector<uint8_t> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)v.size());
}

int main(int argc, char **argv)
{
    try {
        if (argc < 5) throw std::runtime_error("usage: see the header of node_sim.cpp");
        const std::string mode = argv[1];
        const int w = atoi(argv[3]), h = atoi(argv[4]);
        const std::vector<uint8_t> frame = slurp(argv[2], (size_t)w * h * 3);
        if (mode == "pose") {
            if (argc != 18) throw std::runtime_error("pose: 17 arguments expected");
            imgproc_node::PoseEstimateShim node(0);
            imgproc_node::HsvThreshConfig cfg;
            cfg.H_MIN = atoi(argv[11]); cfg.S_MIN = atoi(argv[12]); cfg.V_MIN = atoi(argv[13]);
            cfg.H_MAX = atoi(argv[14]); cfg.S_MAX = atoi(argv[15]); cfg.V_MAX = atoi(argv[16]);
            node.dynamicReconfigureCallback(cfg);
            imgproc_node::RegionOfInterest cam, obj;
            cam.x_offset = (uint32_t)atoi(argv[9]); cam.y_offset = (uint32_t)atoi(argv[10]);
            node.roiCb(cam);
            obj.x_offset = (uint32_t)atoi(argv[5]); obj.y_offset = (uint32_t)atoi(argv[6]);
            obj.width = (uint32_t)atoi(argv[7]); obj.height = (uint32_t)atoi(argv[8]);
            node.objectRoiCb(obj);
            auto o = node.imageCb(frame.data(), (size_t)w * 3, w, h);
            dump(argv[17], o.mask);
            printf("{\"published\": %s, \"position_x\": %.17g, \"position_y\": %.17g, \"center_x\": %d, \"center_y\": %d, "
                   "\"m00\": %llu, \"offsetX\": %d, \"offsetY\": %d, \"mask_w\": %d, \"mask_h\": %d, \"H_MIN\": %d, \"H_MAX\": %d}\n",
                   o.published ? "true" : "false", o.position_x, o.position_y, o.center_x, o.center_y,
                   (unsigned long long)o.m00, node.offsetX, node.offsetY, o.mask_w, o.mask_h, node.H_MIN, node.H_MAX);
        } else if (mode == "polygon") {
            if (argc != 13) throw std::runtime_error("polygon: 12 arguments expected");
            imgproc_node::PolygonPathShim node(0);
            node.H_MIN = atoi(argv[5]); node.S_MIN = atoi(argv[6]); node.V_MIN = atoi(argv[7]);
            node.H_MAX = atoi(argv[8]); node.S_MAX = atoi(argv[9]); node.V_MAX = atoi(argv[10]);
            auto o = node.imageCb(frame.data(), (size_t)w * 3, w, h);
            dump(argv[11], o.thresh);
            dump(argv[12], o.proc);
            printf("{\"m00\": %llu, \"m10\": %llu, \"m01\": %llu, \"cx\": %.17g, \"cy\": %.17g, \"found\": %d}\n",
                   (unsigned long long)o.moments.m00, (unsigned long long)o.moments.m10, (unsigned long long)o.moments.m01,
                   o.moments.cx, o.moments.cy, o.moments.found);
        } else
            throw std::runtime_error("mode must be pose or polygon");
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "node_sim: %s\n", e.what());
        return 1;
    }
}
