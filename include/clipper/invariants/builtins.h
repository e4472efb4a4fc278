/** @file builtins.h — convenience include of the built-in invariants (reference builtins.h). */
#pragma once

#include "clipper/invariants/euclidean_distance.h"
#include "clipper/invariants/pointnormal_distance.h"
