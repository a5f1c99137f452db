#include "opencv2/highgui.hpp"
