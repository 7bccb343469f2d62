// TEST INFRASTRUCTURE: stands in for <torch/torch.h> when the reference's .cu files are compiled as host C++ (oracle/ref_build.py)
#pragma once
#include "../cuda_host_shim.h"
