// SYNTAX-CHECK MOCK, NOT JOLT: see Jolt/Jolt.h of this directory.
#pragma once
#include <Jolt/Jolt.h>
