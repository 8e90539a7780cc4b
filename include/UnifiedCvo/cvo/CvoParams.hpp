// cvo::CvoParams and the yaml surface of the reference (include/UnifiedCvo/cvo/CvoParams.hpp upstream).
// The struct IS the C-ABI's cvo_params_t (same members, order and types as the upstream struct), so a
// CvoParams can be handed to the backend unchanged.
#pragma once
#include <string>
#include <vector>

#include "cvo_hip.h"

namespace cvo {

struct CvoParams : public cvo_params_t {
  CvoParams() { cvo_params_default(this); }
};

// Every key optional, unknown keys ignored, defaults kept otherwise (upstream CvoParams.hpp:193-303).
// Tolerates the quirks of the shipped files: "%YAML:1.0", comments, duplicate keys (first wins) and
// unresolved git conflict markers (HEAD side wins).  Throws std::runtime_error if the file cannot be
// opened or a recognised key has an unparsable value.  Notes about quirks are appended to *warnings.
void read_CvoParams_yaml(const char* filename, CvoParams* params, std::vector<std::string>* warnings = nullptr);
void parse_CvoParams_yaml_text(const std::string& text, CvoParams* params, std::vector<std::string>* warnings = nullptr);

}  // namespace cvo
