#include "boost/serialization/nvp.hpp"
