// pt_internal.h — shared by the translation units of libpt_amd.so (not part of the ABI).
#pragma once
int pt_fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));  // sets pt_last_error(), returns -1
