// Counterpart of the cv::FileStorage the reference driver reads its settings from
// (event_camera_calib/test/eventCameraCalib.cpp:114-121,150-153,168,204-207; parameters.hpp:15-43;
// CirclesEventFrame.cpp:42-48): the subset of YAML 1.0 such a settings file uses — a `%YAML:1.0` directive, one
// `key: value` mapping per line at top level (keys may contain dots: "Camera.width"), `#` comments, scalars (integers,
// reals in decimal or exponent form, quoted or bare strings) and flow sequences of scalars `[ a, b, c ]`.
// node["key"] >> variable  leaves the variable untouched when the key is absent, as cv::FileNode does for an empty node;
// isNone() tells whether a key exists (eventCameraCalib.cpp:150: EndTime is optional).
#ifndef ECAL_HOST_FILE_SETTINGS_HPP_
#define ECAL_HOST_FILE_SETTINGS_HPP_

#include <cstdlib>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace opengv2 {

class FileNode {
public:
    FileNode() : present_(false) {}
    explicit FileNode(const std::string &raw) : present_(true), raw_(raw) {}
    bool isNone() const { return !present_; }
    bool empty() const { return !present_; }
    operator double() const { return present_ ? std::strtod(first().c_str(), nullptr) : 0.0; }
    operator float() const { return (float) (double) *this; }
    operator int() const { return present_ ? (int) std::strtol(first().c_str(), nullptr, 10) : 0; }
    operator std::string() const { return present_ ? first() : std::string(); }
    // flow sequence `[ a, b, c ]` (a scalar reads as a sequence of one)
    std::vector<double> sequence() const {
        std::vector<double> out;
        if (!present_) return out;
        std::string s = raw_;
        if (!s.empty() && s.front() == '[') s = s.substr(1, s.rfind(']') == std::string::npos ? std::string::npos : s.rfind(']') - 1);
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            const std::string tok = trim(s.substr(i, j - i));
            if (!tok.empty()) out.push_back(std::strtod(tok.c_str(), nullptr));
            i = j + 1;
        }
        return out;
    }
    static std::string trim(const std::string &s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }

private:
    std::string first() const {
        std::string s = raw_;
        if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) s = s.substr(1, s.size() - 2);
        return s;
    }
    bool present_;
    std::string raw_;
};

inline void operator>>(const FileNode &n, double &v) { if (!n.isNone()) v = (double) n; }
inline void operator>>(const FileNode &n, float &v) { if (!n.isNone()) v = (float) n; }
inline void operator>>(const FileNode &n, int &v) { if (!n.isNone()) v = (int) n; }
inline void operator>>(const FileNode &n, bool &v) { if (!n.isNone()) v = (int) n != 0; }   // cv reads bool through int
inline void operator>>(const FileNode &n, std::string &v) { if (!n.isNone()) v = (std::string) n; }
inline void operator>>(const FileNode &n, std::vector<double> &v) { if (!n.isNone()) v = n.sequence(); }

class FileSettings {
public:
    FileSettings() : opened_(false) {}
    explicit FileSettings(const std::string &path) : opened_(false) { open(path); }
    bool open(const std::string &path) {
        std::ifstream f(path);
        if (!f) return opened_ = false;
        std::string line;
        while (std::getline(f, line)) {
            bool in_quote = false;   // strip a comment that is not inside a quoted string
            for (size_t i = 0; i < line.size(); i++) {
                if (line[i] == '"') in_quote = !in_quote;
                if (line[i] == '#' && !in_quote) {
                    line.resize(i);
                    break;
                }
            }
            const std::string t = FileNode::trim(line);
            if (t.empty() || t[0] == '%' || t == "---" || t == "...") continue;
            const size_t c = t.find(':');
            if (c == std::string::npos) throw std::runtime_error("settings file: cannot parse line '" + t + "'");
            values_[FileNode::trim(t.substr(0, c))] = FileNode::trim(t.substr(c + 1));
        }
        return opened_ = true;
    }
    bool isOpened() const { return opened_; }
    FileNode operator[](const std::string &key) const {
        auto it = values_.find(key);
        return it == values_.end() ? FileNode() : FileNode(it->second);
    }

private:
    bool opened_;
    std::map<std::string, std::string> values_;
};

}  // namespace opengv2

#endif
