// Forwarding header: same include path as the reference, one facade implementation.
#pragma once
#include "../../fedd_facade.hpp"
