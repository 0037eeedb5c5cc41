// Test harness around the REFERENCE decoder (not a copy of it): exposes CPhoneRecog::Init / RecogOne of
// /root/reference/recog_src/source/phone_recog.cpp through a C ABI so tests/golden/make_viterbi_golden.py can run the
// real reference algorithm on synthetic decoding graphs.  Built by oracle/Makefile from the reference sources where they
// lie (phone_recog.cpp, data.cpp, mlf_io.cpp); the reference's main.cpp (which needs the absent my_stopwatch.h) is not
// part of this library.  TEST INFRASTRUCTURE ONLY.
#include <string.h>
#include <vector>
#include "phone_recog.h"

extern "C" {

void* ref_recog_create(const char* img_file, float PIP, float LMW)
{
    CPhoneRecog* r = new CPhoneRecog();
    if (r->Init((char*)img_file, PIP, LMW) != RETURN_OK) { delete r; return NULL; }
    return r;
}

// lp: [rows, cols] row-major float32.  Writes up to max_seg segments; phones are copied into `phones` as NUL-separated
// strings (phones_cap bytes).  Returns the number of segments, or -1 on failure / overflow.
int ref_recog_one(void* handle, float* lp, int rows, int cols, int* begins, int* ends, char* phones, int phones_cap, int max_seg)
{
    CPhoneRecog* r = (CPhoneRecog*)handle;
    CData d;
    d.ImportByRef(lp, rows, cols);
    std::vector<sResult> res;
    int rc = r->RecogOne(&d, res);
    d.data = NULL;  // ImportByRef does not own the buffer
    d.numRows = d.numColumns = 0;
    if (rc != RETURN_OK || (int)res.size() > max_seg) return -1;
    int pos = 0;
    for (size_t i = 0; i < res.size(); ++i) {
        begins[i] = res[i].begin;
        ends[i] = res[i].end;
        int len = (int)strlen(res[i].phone) + 1;
        if (pos + len > phones_cap) return -1;
        memcpy(phones + pos, res[i].phone, len);
        pos += len;
        delete[] res[i].phone;
    }
    return (int)res.size();
}

void ref_recog_destroy(void* handle) { delete (CPhoneRecog*)handle; }

}
