// svo_ini.cpp -- loadParamsFromConfigFile (libstereo-odometry/include/libstereo-odometry.h:551-672) for the flat svo_params
// record: the reference reads its seven parameter groups from seven named sections of an MRPT INI file; this is the same
// key list, the same "keep the current value when the key is absent" rule and the same quirk (`if_match_method` falls back
// to 0, not to the current value, H:611).  Keys of groups that are not on the path (KLT, SAD, GUI, file output) are accepted
// and ignored, as a file written for the reference carries them.
//
// File grammar = what mrpt::utils::CConfigFile (MRPT 1.x, a SimpleIni front end; not vendored in the reference, not in this
// image) accepts for such files: `[section]` headers, `key = value` lines, whole-line comments starting with ';' or '#',
// trailing `// comment` after a value when preceded by white space, section and key names case-insensitive, a later duplicate
// of a key overriding an earlier one.  read_bool takes true / false / yes / no or an integer.  Host code only; no HIP.
#include "../../include/svo_hip.h"
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

namespace {

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
std::string lower(std::string s) { for (auto& ch : s) ch = (char)tolower((unsigned char)ch); return s; }

typedef std::map<std::string, std::map<std::string, std::string> > Ini;

bool parse_ini(const char* path, Ini& ini)
{
    FILE* f = fopen(path, "r");
    if (!f) return false;
    std::string section, line;
    int ch;
    auto flush = [&]() {
        std::string t = trim(line);
        line.clear();
        if (t.empty() || t[0] == ';' || t[0] == '#') return;
        if (t[0] == '[') { const size_t e = t.find(']'); if (e != std::string::npos) section = lower(trim(t.substr(1, e - 1))); return; }
        const size_t eq = t.find('=');
        if (eq == std::string::npos) return;
        std::string val = trim(t.substr(eq + 1));
        const size_t cpos = val.find("//");
        if (cpos != std::string::npos && cpos > 0 && isspace((unsigned char)val[cpos - 1])) val = trim(val.substr(0, cpos));
        ini[section][lower(trim(t.substr(0, eq)))] = val;
    };
    while ((ch = fgetc(f)) != EOF) { if (ch == '\n') flush(); else if (ch != '\r') line.push_back((char)ch); }
    flush();
    fclose(f);
    return true;
}

struct Reader {
    const Ini& ini; std::string sec;
    const std::string* find(const char* key) const {
        auto s = ini.find(sec);
        if (s == ini.end()) return nullptr;
        auto k = s->second.find(lower(key));
        return k == s->second.end() ? nullptr : &k->second;
    }
    int read_int(const char* key, int def) const { const std::string* v = find(key); return v ? atoi(v->c_str()) : def; }
    double read_double(const char* key, double def) const { const std::string* v = find(key); return v ? atof(v->c_str()) : def; }
    int read_bool(const char* key, int def) const {
        const std::string* v = find(key);
        if (!v) return def;
        const std::string s = lower(trim(*v));
        if (s == "true" || s == "yes") return 1;
        if (s == "false" || s == "no") return 0;
        return atoi(s.c_str()) != 0;
    }
};

}  // namespace

extern "C" int svo_params_load_ini(const char* path, const char* const sections[7], svo_params* p)
{
    if (!path || !sections || !p) return SVO_ERR_ARG;
    Ini ini;
    if (!parse_ini(path, ini)) return SVO_ERR_ARG;                                              // ASSERT_(fileExists), H:669
    auto on = [&](int i) { return sections[i] && sections[i][0]; };
    if (on(0)) {                                                                               // RECTIFY   H:558-559
        Reader r{ ini, lower(sections[0]) };
        p->nOctaves = r.read_int("nOctaves", p->nOctaves);
    }
    if (on(1)) {                                                                               // DETECT    H:561-587
        Reader r{ ini, lower(sections[1]) };
        p->detect_method = r.read_int("detect_method", p->detect_method);
        p->min_distance = r.read_int("min_distance", p->min_distance);
        p->initial_FAST_threshold = r.read_int("initial_FAST_threshold", p->initial_FAST_threshold);
        p->fast_min_th = r.read_int("fast_min_th", p->fast_min_th);
        p->fast_max_th = r.read_int("fast_max_th", p->fast_max_th);
        p->orb_nfeats = r.read_int("orb_nfeats", p->orb_nfeats);
        p->orb_nlevels = r.read_int("orb_nlevels", p->orb_nlevels);
        p->minimum_ORB_response = r.read_double("minimum_ORB_response", p->minimum_ORB_response);
        p->non_maximal_suppression = r.read_bool("non_maximal_suppression", p->non_maximal_suppression);
        p->nmsMethod = r.read_int("non_max_supp_method", p->nmsMethod);
    }
    if (on(2)) {                                                                               // MATCH     H:589-605
        Reader r{ ini, lower(sections[2]) };
        p->match_method = r.read_int("match_method", p->match_method);
        p->max_y_diff = r.read_double("max_y_diff", p->max_y_diff);
        p->enable_robust_1to1_match = r.read_bool("enable_robust_1to1_match", p->enable_robust_1to1_match);
        p->orb_min_th = r.read_int("orb_min_th", p->orb_min_th);
        p->orb_max_th = r.read_int("orb_max_th", p->orb_max_th);
        p->orb_max_distance = r.read_double("orb_max_distance", p->orb_max_distance);
    }
    if (on(3)) {                                                                               // IF-MATCH  H:607-623
        Reader r{ ini, lower(sections[3]) };
        p->ifm_method = r.read_int("if_match_method", 0);                                      // default 0, not the current value (H:611)
        p->filter_fund_matrix = r.read_bool("filter_fund_matrix", p->filter_fund_matrix);
        p->ifm_win_h = r.read_int("window_height", p->ifm_win_h);
        p->ifm_win_w = r.read_int("window_width", p->ifm_win_w);
        // params_if_match.orb_max_distance (H:622) is "unused by now" (H:301): no field
    }
    if (on(4)) {                                                                               // LEAST_SQUARES H:625-643
        Reader r{ ini, lower(sections[4]) };
        p->use_previous_pose_as_initial = r.read_bool("use_previous_pose_as_initial", p->use_previous_pose_as_initial);
        p->initial_max_iters = r.read_int("initial_max_iters", p->initial_max_iters);
        p->max_iters = r.read_int("max_iters", p->max_iters);
        p->min_mod_out_vector = r.read_double("min_mod_out_vector", p->min_mod_out_vector);
        p->max_incr_cost = r.read_int("max_incr_cost", p->max_incr_cost);
        p->residual_threshold = r.read_double("residual_threshold", p->residual_threshold);
        p->bad_tracking_th = r.read_int("bad_tracking_th", p->bad_tracking_th);
        p->use_robust_kernel = r.read_bool("use_robust_kernel", p->use_robust_kernel);
        p->kernel_param = r.read_double("kernel_param", p->kernel_param);
    }
    // sections[5] (GUI, H:645-651): nothing on the path
    if (on(6)) {                                                                               // GENERAL   H:653-660
        Reader r{ ini, lower(sections[6]) };
        p->vo_use_matches_ids = r.read_bool("vo_use_matches_ids", p->vo_use_matches_ids);
    }
    return SVO_OK;
}
