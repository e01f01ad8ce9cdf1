// extern "C" entry points of include/tedd.h over Ctx<TEDD_BACKEND>.  Included once by the
// translation unit that defines TEDD_BACKEND (tedd_hip.hip for the shipped library).
#pragma once
#include <cstring>
#include <new>

#include "tedd_ctx.hpp"

#ifndef TEDD_BACKEND
#error "define TEDD_BACKEND before including tedd_api_impl.hpp"
#endif

struct tedd_ctx {
    tedd::Ctx<TEDD_BACKEND>* impl;
    std::string err;
};

namespace {
thread_local std::string g_create_error;

template <class Fn>
int guarded(tedd_ctx* c, Fn&& fn) {
    if (!c || !c->impl) return TEDD_ERR_INVALID;
    typename TEDD_BACKEND::DeviceKeeper keep;
    try {
        c->impl->be.use_device();
        try {
            fn(*c->impl);
        } catch (...) {
            // launches recorded for the persistent step kernel before the call failed still run: the steps
            // completed so far stand, as they do with separate launches
            try {
                c->impl->be.rec_end();
            } catch (...) {
                c->impl->be.rec_discard();
            }
            throw;
        }
        c->impl->be.rec_end();
        return TEDD_OK;
    } catch (const tedd::AssertError& e) {
        c->err = e.what();
        return TEDD_ERR_ASSERT;
    } catch (const tedd::InvalidError& e) {
        c->err = e.what();
        return TEDD_ERR_INVALID;
    } catch (const tedd::NotConverged& e) {
        c->err = e.what();
        return TEDD_ERR_NOT_CONVERGED;
    } catch (const std::exception& e) {
        c->err = e.what();
        return TEDD_ERR_RUNTIME;
    }
}
}  // namespace

extern "C" {

int tedd_create(tedd_ctx** out, const tedd_config* cfg, const double* x, const double* y) {
    if (!out || !cfg || !x || !y) {
        g_create_error = "null argument";
        return TEDD_ERR_INVALID;
    }
    *out = nullptr;
    typename TEDD_BACKEND::DeviceKeeper keep;
    try {
        tedd_ctx* c = new tedd_ctx();
        try {
            c->impl = new tedd::Ctx<TEDD_BACKEND>(*cfg, x, y);
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
        return TEDD_OK;
    } catch (const tedd::InvalidError& e) {
        g_create_error = e.what();
        return TEDD_ERR_INVALID;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return TEDD_ERR_RUNTIME;
    }
}

void tedd_destroy(tedd_ctx* ctx) {
    if (!ctx) return;
    typename TEDD_BACKEND::DeviceKeeper keep;
    try {
        if (ctx->impl) ctx->impl->be.use_device();
        delete ctx->impl;
    } catch (...) {
    }
    delete ctx;
}

const char* tedd_last_error(const tedd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int tedd_set_model(tedd_ctx* ctx, const double consts[17], int32_t model_kind, double eta, const double mms_params[8]) {
    return guarded(ctx, [&](auto& C) {
        if (!consts) throw tedd::InvalidError("consts is null");
        C.set_model(consts, model_kind, eta, mms_params);
    });
}
int tedd_set_mms_program(tedd_ctx* ctx, int32_t which, const uint32_t* code, int32_t n_ops, const double* consts,
                         int32_t n_consts, int32_t n_regs) {
    return guarded(ctx, [&](auto& C) { C.set_mms_program(which, code, n_ops, consts, n_consts, n_regs); });
}
int tedd_set_option(tedd_ctx* ctx, const char* name, double value) {
    return guarded(ctx, [&](auto& C) {
        if (!name) throw tedd::InvalidError("name is null");
        C.set_option(name, value);
    });
}
int tedd_set_stream(tedd_ctx* ctx, void* s) {
    return guarded(ctx, [&](auto& C) { C.be.set_stream(s); });
}
int tedd_sync(tedd_ctx* ctx) {
    return guarded(ctx, [&](auto& C) { C.be.sync(); });
}
int32_t tedd_local_rows(const tedd_ctx* ctx) { return ctx && ctx->impl ? ctx->impl->nrows : -1; }
int32_t tedd_row_offset(const tedd_ctx* ctx) { return ctx && ctx->impl ? ctx->impl->i0 : -1; }
int32_t tedd_pitch(const tedd_ctx* ctx) { return ctx && ctx->impl ? ctx->impl->ld : -1; }
const char* tedd_placement(const tedd_ctx* ctx) { return ctx && ctx->impl ? ctx->impl->placement() : ""; }

int tedd_set_state(tedd_ctx* ctx, const double* cp, const double* T, const double* cl, const double* cd, const double* cs) {
    return guarded(ctx, [&](auto& C) {
        const double* h[5] = {cp, T, cl, cd, cs};
        for (auto p : h)
            if (!p) throw tedd::InvalidError("null state array");
        C.set_state(h);
    });
}
int tedd_get_state(tedd_ctx* ctx, double* cp, double* T, double* cl, double* cd, double* cs) {
    return guarded(ctx, [&](auto& C) {
        double* h[5] = {cp, T, cl, cd, cs};
        C.get_state(h);
    });
}
int tedd_state_device_ptrs(tedd_ctx* ctx, void* ptrs[5]) {
    return guarded(ctx, [&](auto& C) {
        if (!ptrs) throw tedd::InvalidError("ptrs is null");
        C.require_state();
        for (int v = 0; v < 5; ++v) ptrs[v] = C.cur[v];
    });
}
int tedd_record_device_ptr(tedd_ctx* ctx, void** ptr, int32_t* n) {
    return guarded(ctx, [&](auto& C) {
        if (!ptr || !n) throw tedd::InvalidError("null output pointer");
        *ptr = C.d_red;
        *n = 16;
    });
}
int tedd_prefetch_forcing(tedd_ctx* ctx, double t, double keep_t) {
    return guarded(ctx, [&](auto& C) { C.prefetch_forcing(t, keep_t); });
}
int tedd_set_ghost_hook(tedd_ctx* ctx, tedd_ghost_fn fn, void* user) {
    return guarded(ctx, [&](auto& C) {
        C.ghost_hook = fn;
        C.ghost_user = user;
    });
}
int tedd_ghosts_pending(tedd_ctx* ctx) {
    return guarded(ctx, [&](auto& C) {
        C.require_state();
        if (C.gl == 0 && C.gr == 0) throw tedd::InvalidError("a context that holds the whole mesh has no ghost rows");
        C.ghost_pending = true;
    });
}
int tedd_ghosts_refreshed(tedd_ctx* ctx) {
    return guarded(ctx, [&](auto& C) { C.mark_state_exchanged(); });
}
int tedd_set_state_exact(tedd_ctx* ctx, double t) {
    return guarded(ctx, [&](auto& C) { C.set_state_exact(t); });
}
int tedd_set_forcing(tedd_ctx* ctx, int32_t which, const double* fcp, const double* fT, const double* fcl,
                     const double* fcd, const double* fcs) {
    return guarded(ctx, [&](auto& C) {
        const double* h[5] = {fcp, fT, fcl, fcd, fcs};
        for (auto p : h)
            if (!p) throw tedd::InvalidError("null forcing array");
        C.set_forcing(which, h);
    });
}
int tedd_step_pc(tedd_ctx* ctx, double t0, double dt, int32_t num_pc, int32_t num_newton, int32_t cs_iters,
                 double cs_rtol, const int32_t jacobi_sweeps[3], double jacobi_tol, int32_t jacobi_max) {
    return guarded(ctx, [&](auto& C) { C.step_pc(t0, dt, num_pc, num_newton, cs_iters, cs_rtol, jacobi_sweeps, jacobi_tol, jacobi_max); });
}
int tedd_step_feuler(tedd_ctx* ctx, double t0, double dt) {
    return guarded(ctx, [&](auto& C) {
        if (!(dt > 0)) throw tedd::AssertError("dt must be > 0");
        C.step_feuler(t0, dt);
    });
}
int tedd_advance(tedd_ctx* ctx, double t0, double dt, int32_t nsteps, int32_t integrator, int32_t num_pc,
                 int32_t num_newton, int32_t cs_iters, double cs_rtol, const int32_t jacobi_sweeps[3],
                 double jacobi_tol, int32_t jacobi_max, double* err_series) {
    return guarded(ctx, [&](auto& C) {
        C.advance(t0, dt, nsteps, integrator, num_pc, num_newton, cs_iters, cs_rtol, jacobi_sweeps, jacobi_tol,
                  jacobi_max, err_series);
    });
}
int tedd_predict_cp_cs(tedd_ctx* ctx, double t, double dt, double* cp1, double* cs1) {
    return guarded(ctx, [&](auto& C) {
        const double* in[3] = {nullptr, nullptr, nullptr};
        double* out[2] = {cp1, cs1};
        C.substep(0, t, dt, in, out, 0, 0.0, 0, 0.0, 0);
    });
}
int tedd_correct_cp_cs(tedd_ctx* ctx, double t0, double dt, const double* T1, const double* cl1, const double* cd1,
                       int32_t cs_newton_iterations, double cs_consec_xs_rtol, double* cp1, double* cs1) {
    return guarded(ctx, [&](auto& C) {
        const double* in[3] = {T1, cl1, cd1};
        double* out[2] = {cp1, cs1};
        C.substep(1, t0, dt, in, out, cs_newton_iterations, cs_consec_xs_rtol, 0, 0.0, 0);
    });
}
int tedd_newton_step(tedd_ctx* ctx, int32_t var, double t0, double dt, const double* Y, const double* T1,
                     const double* cl1, int32_t jacobi_sweeps, double jacobi_tol, int32_t jacobi_max, double* out,
                     double* residual) {
    return guarded(ctx, [&](auto& C) {
        if (var < 1 || var > 3) throw tedd::InvalidError("var must be 1 (T), 2 (cl) or 3 (cd)");
        const double* in[3] = {Y, T1, cl1};
        double* o[2] = {out, residual};
        C.substep(1 + var, t0, dt, in, o, 0, 0.0, jacobi_sweeps, jacobi_tol, jacobi_max);
    });
}
int tedd_get_residuals(tedd_ctx* ctx, double* rT, double* rcl, double* rcd) {
    return guarded(ctx, [&](auto& C) {
        double* h[3] = {rT, rcl, rcd};
        C.get_residuals(h);
    });
}
int tedd_eval_field(tedd_ctx* ctx, double t, double* Fcp, double* FT, double* Fcl, double* Fcd, double* Fcs) {
    return guarded(ctx, [&](auto& C) {
        double* h[5] = {Fcp, FT, Fcl, Fcd, Fcs};
        C.eval_field(t, h);
    });
}
int tedd_eval_exact(tedd_ctx* ctx, double t, double* cp, double* T, double* cl, double* cd, double* cs) {
    return guarded(ctx, [&](auto& C) {
        double* h[5] = {cp, T, cl, cd, cs};
        C.eval_exact(t, h);
    });
}
int tedd_eval_mms(tedd_ctx* ctx, double t, int32_t what, double* cp, double* T, double* cl, double* cd, double* cs) {
    return guarded(ctx, [&](auto& C) {
        double* h[5] = {cp, T, cl, cd, cs};
        C.eval_exact(t, h, what);
    });
}
int tedd_eval_forcing(tedd_ctx* ctx, double t, double* fcp, double* fT, double* fcl, double* fcd, double* fcs) {
    return guarded(ctx, [&](auto& C) {
        double* h[5] = {fcp, fT, fcl, fcd, fcs};
        C.eval_forcing(t, h);
    });
}
int tedd_eval_jacobian(tedd_ctx* ctx, int32_t op, double* c, double* e, double* w, double* n, double* s) {
    return guarded(ctx, [&](auto& C) {
        double* h[5] = {c, e, w, n, s};
        C.eval_jacobian(op, h);
    });
}
int tedd_error_norms(tedd_ctx* ctx, double t, const double* const exact[5], double out[8]) {
    return guarded(ctx, [&](auto& C) {
        if (!out) throw tedd::InvalidError("out is null");
        C.error_norms(t, exact, out);
    });
}
int tedd_error_norms_noalloc(tedd_ctx* ctx, double t, double* out_device) {
    return guarded(ctx, [&](auto& C) { C.error_norms_noalloc(t, out_device); });
}
int tedd_get_stats(tedd_ctx* ctx, tedd_step_stats* out) {
    return guarded(ctx, [&](auto& C) {
        if (!out) throw tedd::InvalidError("out is null");
        C.get_stats(out);
    });
}
int tedd_check_converged(tedd_ctx* ctx) {
    return guarded(ctx, [&](auto& C) { C.check_converged(); });
}
int tedd_set_reduce_hook(tedd_ctx* ctx, tedd_reduce_fn fn, void* user) {
    return guarded(ctx, [&](auto& C) { C.set_reduce_hook(fn, user); });
}
int tedd_profile_enable(tedd_ctx* ctx, int32_t enable) {
    return guarded(ctx, [&](auto& C) { C.be.profile_enable(enable != 0); });
}
int tedd_profile_get(tedd_ctx* ctx, int32_t cap, const char** names, int64_t* launches, double* total_ms) {
    int n = 0;
    const int rc = guarded(ctx, [&](auto& C) {
        if (cap < 0 || (cap > 0 && (!names || !launches || !total_ms))) throw tedd::InvalidError("null output array");
        C.be.profile_collect();
        for (int f = 0; f < tedd::FAM_COUNT && n < cap; ++f) {
            if (C.be.fam_n[f] == 0) continue;
            names[n] = tedd::fam_name(f);
            launches[n] = C.be.fam_n[f];
            total_ms[n] = C.be.fam_ms[f];
            ++n;
        }
    });
    return rc < 0 ? rc : n;
}

}  // extern "C"
