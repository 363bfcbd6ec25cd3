// scratch_pool.cpp -- see scratch_pool.hpp
#include "scratch_pool.hpp"

#include <mutex>

namespace sdrhip {

namespace {
struct Pool {
    std::mutex mu;
    std::vector<ScratchCtx*> idle;
};
Pool& pool()
{
    static Pool* p = new Pool();
    return *p;
}
}  // namespace

ScratchCtx* scratch_acquire()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error("hipGetDevice failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    Pool& p = pool();
    {
        std::lock_guard<std::mutex> lk(p.mu);
        for (size_t i = 0; i < p.idle.size(); i++)
            if (p.idle[i]->device == dev) {
                ScratchCtx* c = p.idle[i];
                p.idle[i] = p.idle.back();
                p.idle.pop_back();
                return c;
            }
    }
    ScratchCtx* c = new ScratchCtx();
    c->device = dev;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
        delete c;
        return nullptr;
    }
    return c;
}

void scratch_release(ScratchCtx* c)
{
    Pool& p = pool();
    std::lock_guard<std::mutex> lk(p.mu);
    p.idle.push_back(c);
}

}  // namespace sdrhip
