#include "CmdLine.h"
