'use strict';
// Loads the N-API addon (pragma-dsp_amd/csrc/pdsp_napi.node -> libpdsp_hip.so).
// There is no JS fallback: if the addon is missing this throws at require time.
const path = require('path');
module.exports = require(path.join(__dirname, '..', 'csrc', 'pdsp_napi.node'));
