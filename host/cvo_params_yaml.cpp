// Tolerant reader of the cvo_params/*.yaml surface (see include/UnifiedCvo/cvo/CvoParams.hpp).
#include <cstddef>
#include <cstdlib>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>

#include "cvo/CvoParams.hpp"

namespace cvo {
namespace {

enum Kind { F32, F64, I32 };
struct Key {
  const char* name;
  Kind kind;
  size_t offset;
};
#define K_F(n) {#n, F32, offsetof(cvo_params_t, n)}
#define K_D(n) {#n, F64, offsetof(cvo_params_t, n)}
#define K_I(n) {#n, I32, offsetof(cvo_params_t, n)}
// exactly the keys upstream's reader asks for
const Key kKeys[] = {
    K_F(ell_init_first_frame), K_F(ell_init), K_F(ell_min), K_I(min_ell_iter_limit), K_F(ell_max), K_D(dl),
    K_D(dl_step), K_F(sigma), K_F(sp_thres), K_F(c), K_F(d), K_F(c_ell), K_F(c_sigma), K_F(s_ell), K_F(s_sigma),
    K_I(MAX_ITER), K_F(eps), K_F(eps_2), K_F(min_step), K_F(max_step), K_F(ell_decay_rate),
    K_F(ell_decay_rate_first_frame), K_I(ell_decay_start), K_I(ell_decay_start_first_frame),
    K_I(indicator_window_size), K_F(indicator_stable_threshold), K_I(is_pcl_visualization_on),
    K_I(is_using_least_square), K_I(is_full_ip_matrix), K_I(is_using_geometry), K_I(is_using_intensity),
    K_I(is_using_semantics), K_I(is_using_range_ell), K_I(is_using_kdtree), K_I(is_using_geometric_type),
    K_I(is_exporting_association), K_I(nearest_neighbors_max), K_I(multiframe_using_cpu),
    K_F(multiframe_ell_init), K_I(multiframe_max_iters), K_F(multiframe_ell_min), K_F(multiframe_ell_decay_rate),
    K_I(multiframe_iterations_per_ell), K_I(multiframe_iterations_per_solve),
    K_F(multiframe_downsample_voxel_size), K_I(multiframe_expected_points), K_I(multiframe_num_neighbors),
    K_I(multiframe_min_nonzeros), K_I(multiframe_least_squares_num_threads),
};

std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

}  // namespace

void parse_CvoParams_yaml_text(const std::string& text, CvoParams* params, std::vector<std::string>* warnings) {
  std::istringstream in(text);
  std::string raw;
  std::set<std::string> seen;
  int side = 0;  // 0 outside a conflict block, 1 = HEAD side, 2 = other side
  int lineno = 0;
  auto warn = [&](const std::string& m) {
    if (warnings) warnings->push_back("line " + std::to_string(lineno) + ": " + m);
  };
  while (std::getline(in, raw)) {
    ++lineno;
    if (raw.rfind("<<<<<<<", 0) == 0) {
      side = 1;
      warn("unresolved git conflict marker; taking the HEAD side");
      continue;
    }
    if (side && raw.rfind("=======", 0) == 0) {
      side = 2;
      continue;
    }
    if (side && raw.rfind(">>>>>>>", 0) == 0) {
      side = 0;
      continue;
    }
    if (side == 2) continue;
    std::string line = raw.substr(0, raw.find('#'));
    line = trim(line);
    if (line.empty() || line[0] == '%' || line == "---" || line == "...") continue;
    size_t colon = line.find(':');
    if (colon == std::string::npos) {
      warn("ignored");
      continue;
    }
    const std::string key = trim(line.substr(0, colon));
    std::string value = trim(line.substr(colon + 1));
    if (value.size() >= 2 && value.front() == value.back() && (value.front() == '"' || value.front() == '\''))
      value = value.substr(1, value.size() - 2);
    const Key* k = nullptr;
    for (const Key& c : kKeys)
      if (key == c.name) k = &c;
    if (!k) continue;  // unknown keys are never asked for
    if (!seen.insert(key).second) {
      warn("duplicate key '" + key + "'; keeping the first value");
      continue;
    }
    char* end = nullptr;
    char* base = reinterpret_cast<char*>(static_cast<cvo_params_t*>(params));
    if (k->kind == I32) {
      long v = std::strtol(value.c_str(), &end, 10);
      if (end == value.c_str() || *end != '\0') throw std::runtime_error("cannot parse " + key + ": '" + value + "'");
      *reinterpret_cast<int*>(base + k->offset) = (int)v;
    } else {
      double v = std::strtod(value.c_str(), &end);
      if (end == value.c_str() || *end != '\0') throw std::runtime_error("cannot parse " + key + ": '" + value + "'");
      if (k->kind == F32)
        *reinterpret_cast<float*>(base + k->offset) = (float)v;
      else
        *reinterpret_cast<double*>(base + k->offset) = v;
    }
  }
}

void read_CvoParams_yaml(const char* filename, CvoParams* params, std::vector<std::string>* warnings) {
  std::ifstream f(filename);
  if (!f) throw std::runtime_error(std::string("cannot open CvoParams yaml file ") + filename);
  std::stringstream ss;
  ss << f.rdbuf();
  parse_CvoParams_yaml_text(ss.str(), params, warnings);
}

}  // namespace cvo
