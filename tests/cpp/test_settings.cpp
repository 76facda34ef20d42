// CPU: the settings reader (host/file_settings.hpp) on a file with the shapes a cv::FileStorage settings file has.
//   usage: test_settings file.yaml   -> prints the parsed values
#include <cstdio>

#include "../../eventcalib_amd/csrc/host/file_settings.hpp"

int main(int argc, char **argv) {
    using namespace opengv2;
    if (argc < 2) return 2;
    FileSettings fs(argv[1]);
    if (!fs.isOpened()) return 3;
    double step = fs["MotionTimeStep"], start = fs["StartTime"];
    int w = fs["Camera.width"], thr = fs["FrameEventNumThreshold"];
    bool so3 = true, missing = true;
    fs["useSO3"] >> so3;
    fs["NoSuchKey"] >> missing;            // absent: the variable keeps its value
    std::string name = fs["Name"];
    std::vector<double> facing;
    fs["Viewer.Facing"] >> facing;
    std::printf("step %.17g start %.17g width %d thr %d so3 %d missing %d end_none %d name [%s] facing %zu %.1f %.1f\n", step, start, w, thr,
                (int) so3, (int) missing, (int) fs["EndTime"].isNone(), name.c_str(), facing.size(), facing.empty() ? 0.0 : facing[0],
                facing.size() > 4 ? facing[4] : 0.0);
    return 0;
}
