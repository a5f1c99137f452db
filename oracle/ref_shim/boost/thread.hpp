#include "boost/thread/mutex.hpp"
