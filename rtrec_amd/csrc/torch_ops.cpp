// rtrec_amd/csrc/torch_ops.cpp -- the PyTorch-ROCm custom ops `torch.ops.rtrec_amd.*`, registered from C++.
//
// One op per entry point of include/rtrec_amd.h, in out-variant style: every buffer is a ROCm tensor owned by the
// caller, outputs and scratch are marked mutable in the schema, scalars are plain ints / floats / bools, and the work is
// enqueued on the current HIP stream of the tensors' device.  The ops only marshal pointers -- all computation is in
// librtrec_amd.so, which this library binds at run time (rtrec_ops_bind: dlopen + dlsym of the C-ABI, so an A/B build
// of the kernels can be selected with RTREC_AMD_LIB like before) -- so they compose with torch streams, the caching
// allocator and torch.distributed without copies.  Round 2 registered the same schemas from Python
// (torch.library.custom_op bodies that called ctypes); a call now goes dispatcher -> this function -> C-ABI.
//
// Built by rtrec_amd/build.py with the host compiler against the torch headers (no device code here).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <c10/util/Optional.h>
#include <torch/library.h>

#include <dlfcn.h>

#include <stdexcept>
#include <string>
#include <vector>
#include <type_traits>

#include "../../include/rtrec_amd.h"

namespace {

struct Abi {
    void *handle = nullptr;
    decltype(&rtrec_amd_last_error) last_error = nullptr;
    decltype(&rtrec_slim_column_sqnorms) column_sqnorms = nullptr;
    decltype(&rtrec_slim_fit_workspace_init) fit_workspace_init = nullptr;
    decltype(&rtrec_slim_gram_matrix) gram_matrix = nullptr;
    decltype(&rtrec_slim_fit_columns_opt) fit_columns_opt = nullptr;
    decltype(&rtrec_slim_score_topk_opt) score_topk_opt = nullptr;
    decltype(&rtrec_slim_score_rows) score_rows = nullptr;
    decltype(&rtrec_slim_merge_topk_strided) merge_topk_strided = nullptr;
    decltype(&rtrec_slim_similar_topk) similar_topk = nullptr;
    decltype(&rtrec_store_decay_device) store_decay_device = nullptr;
    decltype(&rtrec_store_fold_device) store_fold_device = nullptr;
    decltype(&rtrec_slim_fit_sgd_epochs) fit_sgd_epochs = nullptr;
    decltype(&rtrec_slim_first_touch_aux) first_touch_aux = nullptr;
    decltype(&rtrec_slim_dense_fill) dense_fill = nullptr;
    decltype(&rtrec_slim_refine_topk_f64) refine_topk_f64 = nullptr;
    decltype(&rtrec_slim_score_candidates) score_candidates = nullptr;
    decltype(&rtrec_slim_seg_plan) seg_plan = nullptr;
    decltype(&rtrec_slim_seg_fill) seg_fill = nullptr;
    decltype(&rtrec_slim_ordered_sums) ordered_sums = nullptr;
};
Abi g_abi;

template <typename F>
void bind_one(void *h, F &slot, const char *name) {
    slot = reinterpret_cast<F>(dlsym(h, name));
    if (!slot) throw std::runtime_error(std::string("librtrec_amd.so does not export ") + name);
}

const Abi &abi() {
    TORCH_CHECK(g_abi.handle, "rtrec_amd ops are not bound to librtrec_amd.so (rtrec_amd.ops binds them at import)");
    return g_abi;
}

void check(int status, const char *what) {
    if (status == 0) return;
    const char *names[] = {"ok", "invalid argument", "unsupported parameter", "workspace too small", "kernel launch failed"};
    const int k = -status;
    std::string msg = std::string(what) + " failed: " + (k >= 0 && k <= 4 ? names[k] : "error") + " (" + std::to_string(status) + ")";
    if (status == -4 && g_abi.last_error) msg += std::string(" ") + g_abi.last_error();
    TORCH_CHECK(false, msg);
}

// The C-ABI takes raw device pointers: a tensor of another dtype, a strided view or a host tensor in any slot would be a
// silent out-of-bounds access on the GPU, so every pointer handed over is checked against the element type its slot
// declares.  ptr<void> / ptr<unsigned char> slots are byte workspaces: any dtype, still contiguous + device.
template <typename T> struct ElemOf { static bool ok(at::ScalarType) { return true; } };
template <> struct ElemOf<int32_t> { static bool ok(at::ScalarType s) { return s == at::kInt; } };
template <> struct ElemOf<uint32_t> { static bool ok(at::ScalarType s) { return s == at::kInt || s == at::kUInt32; } };
template <> struct ElemOf<int64_t> { static bool ok(at::ScalarType s) { return s == at::kLong; } };
template <> struct ElemOf<float> { static bool ok(at::ScalarType s) { return s == at::kFloat; } };
template <> struct ElemOf<double> { static bool ok(at::ScalarType s) { return s == at::kDouble; } };

template <typename T>
void check_tensor(const at::Tensor &t) {
    TORCH_CHECK(t.is_contiguous(), "rtrec_amd op: tensor argument must be contiguous (got strides of a view)");
    TORCH_CHECK(t.device().is_cuda(), "rtrec_amd op: tensor argument must live on the GPU, got ", t.device());
    TORCH_CHECK(ElemOf<typename std::remove_const<T>::type>::ok(t.scalar_type()), "rtrec_amd op: tensor argument has dtype ",
                t.scalar_type(), ", the slot takes ", sizeof(T) == 8 ? "a 64-bit" : "a 32-bit", " element type");
}
template <> void check_tensor<void>(const at::Tensor &t) {
    TORCH_CHECK(t.is_contiguous() && t.device().is_cuda(), "rtrec_amd op: workspace tensors must be contiguous GPU tensors");
}
template <> void check_tensor<const void>(const at::Tensor &t) { check_tensor<void>(t); }

// strided inputs (merge_topk: views of an exchanged record buffer, their strides travel as arguments): dtype + device only
template <typename T>
T *sptr(const at::Tensor &t) {
    if (t.numel() == 0) return nullptr;
    TORCH_CHECK(t.device().is_cuda(), "rtrec_amd op: tensor argument must live on the GPU, got ", t.device());
    TORCH_CHECK(ElemOf<typename std::remove_const<T>::type>::ok(t.scalar_type()), "rtrec_amd op: tensor argument has dtype ", t.scalar_type());
    return static_cast<T *>(t.data_ptr());
}
template <typename T>
T *sptr(const c10::optional<at::Tensor> &t) { return (t.has_value() && t->defined()) ? sptr<T>(*t) : nullptr; }

template <typename T = void>
T *ptr(const at::Tensor &t) {
    if (t.numel() == 0) return nullptr;
    check_tensor<T>(t);
    return static_cast<T *>(t.data_ptr());
}
template <typename T = void>
T *ptr(const c10::optional<at::Tensor> &t) {
    if (!(t.has_value() && t->defined() && t->numel() > 0)) return nullptr;
    check_tensor<T>(*t);
    return static_cast<T *>(t->data_ptr());
}

void *stream_of(const at::Tensor &t) {
    return static_cast<void *>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

using OT = c10::optional<at::Tensor>;

void column_sqnorms(const at::Tensor &cptr, const at::Tensor &cval, at::Tensor out) {
    check(abi().column_sqnorms(static_cast<int32_t>(out.size(0)), ptr<const int32_t>(cptr), ptr<const float>(cval), ptr<float>(out),
                               stream_of(out)), "rtrec_slim_column_sqnorms");
}

void fit_workspace_init(at::Tensor ws, int64_t n_users, int64_t n_items, int64_t n_slots, int64_t top_features) {
    check(abi().fit_workspace_init(ptr(ws), static_cast<size_t>(ws.numel()), n_users, n_items, n_slots, top_features, stream_of(ws)),
          "rtrec_slim_fit_workspace_init");
}

void gram_matrix(const at::Tensor &cptr, const at::Tensor &crow, const at::Tensor &cval, const at::Tensor &top_items, at::Tensor ws,
                 at::Tensor gram, int64_t n_users, int64_t n_items) {
    check(abi().gram_matrix(n_users, n_items, ptr<const int32_t>(cptr), ptr<const int32_t>(crow), ptr<const float>(cval),
                            ptr<const int32_t>(top_items), static_cast<int32_t>(top_items.size(0)), ptr(ws),
                            static_cast<size_t>(ws.numel()), ptr<double>(gram), stream_of(gram)), "rtrec_slim_gram_matrix");
}

void fit_columns(const at::Tensor &cptr, const at::Tensor &crow, const at::Tensor &cval, const at::Tensor &rptr, const at::Tensor &rcol,
                 const at::Tensor &rval, const at::Tensor &sqn, const at::Tensor &targets, int64_t n_users, int64_t n_items,
                 double l1_reg, double l2_reg, double tol, int64_t max_iter, int64_t seed, bool positive, int64_t top_features,
                 at::Tensor out_items, at::Tensor out_coef, at::Tensor out_count, at::Tensor out_n_iter, int64_t cap, at::Tensor ws,
                 int64_t n_slots, at::Tensor queue, OT trace, const OT &gram, const OT &gram_index, int64_t gram_n, double gram_rel_err,
                 int64_t fast, int64_t kernel, int64_t colwalk_min_rows, int64_t screen_min, int64_t lane_max, OT xty_ws,
                 const OT &col_order, int64_t fold) {
    rtrec_fit_cfg cfg{static_cast<float>(l1_reg), static_cast<float>(l2_reg), static_cast<float>(tol), static_cast<int32_t>(max_iter),
                      static_cast<uint32_t>(seed), positive ? 1 : 0, static_cast<int32_t>(top_features)};
    rtrec_fit_opts o{};
    o.d_trace = ptr<int64_t>(trace);
    o.d_gram = ptr<const double>(gram);
    o.d_gram_index = ptr<const int32_t>(gram_index);
    o.gram_n = o.d_gram ? static_cast<int32_t>(gram_n) : 0;
    o.gram_rel_err = gram_rel_err;
    o.fast = static_cast<int32_t>(fast);
    o.kernel = static_cast<int32_t>(kernel);
    o.colwalk_min_rows = static_cast<int32_t>(colwalk_min_rows);
    o.screen_min = static_cast<int32_t>(screen_min);
    o.lane_max = static_cast<int32_t>(lane_max);
    o.d_xty_ws = ptr(xty_ws);
    o.xty_ws_bytes = (xty_ws.has_value() && xty_ws->defined()) ? static_cast<size_t>(xty_ws->numel()) : 0;
    o.nnz = rcol.size(0);
    o.d_col_order = ptr<const int32_t>(col_order);
    o.fold = static_cast<int32_t>(fold);
    check(abi().fit_columns_opt(n_users, n_items, ptr<const int32_t>(cptr), ptr<const int32_t>(crow), ptr<const float>(cval),
                                ptr<const int32_t>(rptr), ptr<const int32_t>(rcol), ptr<const float>(rval), ptr<const float>(sqn),
                                ptr<const int32_t>(targets), static_cast<int32_t>(targets.size(0)), &cfg, ptr<int32_t>(out_items),
                                ptr<float>(out_coef), ptr<int32_t>(out_count), ptr<int32_t>(out_n_iter), static_cast<int32_t>(cap), ptr(ws),
                                static_cast<size_t>(ws.numel()), static_cast<int32_t>(n_slots), ptr<int32_t>(queue), stream_of(out_items), &o),
          "rtrec_slim_fit_columns_opt");
}

void score_topk(const OT &row_ids, const at::Tensor &xb_ptr, const at::Tensor &xb_col, const at::Tensor &xb_val, int64_t n_rows,
                int64_t n_items, int64_t n_cols, int64_t col_offset, const OT &col_ids, const OT &col_map, int64_t tile_cols,
                int64_t n_tiles, const OT &tile_ptr, const OT &w_col, const OT &w_val, const OT &dense_idx, const OT &dense_val,
                const OT &row_hdr, const OT &col_rank, int64_t top_k, bool filter_interacted, int64_t mode, bool acc_f64, at::Tensor ids,
                at::Tensor scores, OT scores64, at::Tensor aux, at::Tensor count, at::Tensor ws, const OT &fr_map, const OT &fr_col_ids,
                const OT &fr_col_map, const OT &fr_w, const OT &fr_tile_rows, const OT &fr_tile_off, const OT &fr_super_kb,
                const OT &fr_super_tile, const OT &fr_frag_tile, int64_t fr_rows, int64_t fr_tile_cols, int64_t fr_n_tiles,
                int64_t fr_n_frags, int64_t fr_n_super, int64_t fr_buf_bytes, OT fr_scratch, const OT &row_order, int64_t timer,
                int64_t diagnostics, OT rescored, int64_t row_order_grouped, const OT &sg_info, const OT &sg_ptr, const OT &sg_ent,
                const OT &sg_bound, const OT &sg_col_ids, int64_t sg_tile_cols, int64_t sg_n_tiles, int64_t sg_rows, int64_t sg_n_cols,
                const OT &sg_trow_ptr, const OT &sg_trow, OT sg_scratch, OT flagged, int64_t aux_stream) {
    rtrec_score_opts o{};
    o.n_x_rows = static_cast<int32_t>(xb_ptr.size(0)) - 1;
    o.d_fr_map = ptr<const int32_t>(fr_map);
    o.d_fr_col_ids = ptr<const int32_t>(fr_col_ids);
    o.d_fr_col_map = ptr<const int32_t>(fr_col_map);
    o.d_fr_w = ptr<const float>(fr_w);
    o.d_fr_tile_rows = ptr<const uint64_t>(fr_tile_rows);
    o.d_fr_tile_off = ptr<const int32_t>(fr_tile_off);
    o.d_fr_super_kb = ptr<const int32_t>(fr_super_kb);
    o.d_fr_super_tile = ptr<const int32_t>(fr_super_tile);
    o.d_fr_frag_tile = ptr<const int32_t>(fr_frag_tile);
    o.fr_rows = static_cast<int32_t>(fr_rows); o.fr_tile_cols = static_cast<int32_t>(fr_tile_cols);
    o.fr_n_tiles = static_cast<int32_t>(fr_n_tiles); o.fr_n_frags = static_cast<int32_t>(fr_n_frags);
    o.fr_n_super = static_cast<int32_t>(fr_n_super); o.fr_buf_bytes = static_cast<int32_t>(fr_buf_bytes);
    o.d_fr_scratch = ptr(fr_scratch);
    o.fr_scratch_bytes = (fr_scratch.has_value() && fr_scratch->defined()) ? static_cast<size_t>(fr_scratch->numel()) : 0;
    o.d_row_order = ptr<const int32_t>(row_order);
    o.timer = reinterpret_cast<void *>(static_cast<intptr_t>(timer));
    o.diagnostics = static_cast<int32_t>(diagnostics);
    o.d_rescored = ptr<int32_t>(rescored);
    o.row_order_grouped = static_cast<int32_t>(row_order_grouped & 1);      // the schema holds 64 arguments at most: two flags in one
    o.row_order_longest_first = static_cast<int32_t>((row_order_grouped >> 1) & 1);
    o.d_sg_info = ptr<const int32_t>(sg_info);
    o.d_sg_ptr = ptr<const int32_t>(sg_ptr);
    o.d_sg_ent = ptr<const uint32_t>(sg_ent);
    o.sg_nnz = (sg_ent.has_value() && sg_ent->defined()) ? sg_ent->size(0) : 0;
    o.d_sg_bound = ptr<const uint32_t>(sg_bound);
    o.d_sg_col_ids = ptr<const int32_t>(sg_col_ids);
    o.sg_tile_cols = static_cast<int32_t>(sg_tile_cols); o.sg_n_tiles = static_cast<int32_t>(sg_n_tiles);
    o.sg_rows = static_cast<int32_t>(sg_rows); o.sg_n_cols = static_cast<int32_t>(sg_n_cols);
    o.d_sg_trow_ptr = ptr<const int32_t>(sg_trow_ptr);
    o.d_sg_trow = ptr<const int32_t>(sg_trow);
    o.d_sg_scratch = ptr(sg_scratch);
    o.sg_scratch_bytes = (sg_scratch.has_value() && sg_scratch->defined()) ? static_cast<size_t>(sg_scratch->numel()) : 0;
    o.d_flagged = ptr<int32_t>(flagged);
    o.aux_stream = reinterpret_cast<void *>(static_cast<intptr_t>(aux_stream));
    check(abi().score_topk_opt(static_cast<int32_t>(n_rows), ptr<const int32_t>(row_ids), ptr<const int32_t>(xb_ptr),
                               ptr<const int32_t>(xb_col), ptr<const float>(xb_val), static_cast<int32_t>(n_items),
                               static_cast<int32_t>(n_cols), static_cast<int32_t>(col_offset), ptr<const int32_t>(col_ids),
                               ptr<const int32_t>(col_map), static_cast<int32_t>(tile_cols), static_cast<int32_t>(n_tiles),
                               ptr<const int32_t>(tile_ptr), ptr<const uint16_t>(w_col), ptr<const float>(w_val),
                               ptr<const int32_t>(dense_idx), ptr<const float>(dense_val), ptr<const int32_t>(row_hdr),
                               ptr<const int32_t>(col_rank), static_cast<int32_t>(top_k), filter_interacted ? 1 : 0,
                               static_cast<int32_t>(mode), acc_f64 ? 1 : 0, ptr<int32_t>(ids), ptr<float>(scores), ptr<double>(scores64),
                               ptr<uint32_t>(aux), ptr<int32_t>(count), ptr(ws), static_cast<size_t>(ws.numel()), stream_of(ids), &o),
          "rtrec_slim_score_topk_opt");
}

void score_rows(const OT &row_ids, const at::Tensor &xb_ptr, const at::Tensor &xb_col, const at::Tensor &xb_val, int64_t n_rows,
                int64_t n_items, int64_t n_cols, int64_t col_offset, int64_t tile_cols, int64_t n_tiles, const at::Tensor &tile_ptr,
                const at::Tensor &w_col, const at::Tensor &w_val, bool acc_f64, at::Tensor out) {
    check(abi().score_rows(static_cast<int32_t>(n_rows), ptr<const int32_t>(row_ids), ptr<const int32_t>(xb_ptr), ptr<const int32_t>(xb_col),
                           ptr<const float>(xb_val), static_cast<int32_t>(n_items), static_cast<int32_t>(n_cols),
                           static_cast<int32_t>(col_offset), static_cast<int32_t>(tile_cols), static_cast<int32_t>(n_tiles),
                           ptr<const int32_t>(tile_ptr), ptr<const uint16_t>(w_col), ptr<const float>(w_val), acc_f64 ? 1 : 0, ptr(out),
                           out.stride(0), stream_of(out)), "rtrec_slim_score_rows");
}

// in_* are [n_lists, n_rows, top_k] (in_count [n_lists, n_rows]); strided views into one packed all-gather buffer are
// fine as long as the last dimension is contiguous.
void merge_topk(const at::Tensor &in_ids, const at::Tensor &in_scores, const OT &in_scores64, const at::Tensor &in_aux,
                const at::Tensor &in_count, int64_t top_k, at::Tensor out_ids, at::Tensor out_scores, at::Tensor out_count) {
    TORCH_CHECK(in_ids.dim() == 3 && in_ids.stride(2) == 1 && in_ids.strides() == in_scores.strides() &&
                in_ids.strides() == in_aux.strides(), "merge_topk: ids / scores / aux must share strides, last dimension contiguous");
    int64_t s64_0 = 0, s64_1 = 0;
    if (in_scores64.has_value() && in_scores64->defined()) {
        TORCH_CHECK(in_scores64->stride(2) == 1, "merge_topk: scores64 last dimension must be contiguous");
        s64_0 = in_scores64->stride(0); s64_1 = in_scores64->stride(1);
    }
    check(abi().merge_topk_strided(static_cast<int32_t>(in_ids.size(1)), static_cast<int32_t>(in_ids.size(0)), static_cast<int32_t>(top_k),
                                   sptr<const int32_t>(in_ids), sptr<const float>(in_scores), sptr<const double>(in_scores64),
                                   sptr<const uint32_t>(in_aux), sptr<const int32_t>(in_count), in_ids.stride(0), in_ids.stride(1), s64_0,
                                   s64_1, in_count.stride(0), in_count.stride(1), ptr<int32_t>(out_ids), ptr<float>(out_scores),
                                   ptr<int32_t>(out_count), stream_of(out_ids)), "rtrec_slim_merge_topk_strided");
}

void similar_topk(const at::Tensor &queries, const at::Tensor &wc_ptr, const at::Tensor &wc_row, const at::Tensor &wc_val, int64_t top_k,
                  at::Tensor ids, at::Tensor scores, at::Tensor count) {
    check(abi().similar_topk(static_cast<int32_t>(queries.size(0)), ptr<const int32_t>(queries), ptr<const int32_t>(wc_ptr),
                             ptr<const int32_t>(wc_row), ptr<const float>(wc_val), static_cast<int32_t>(top_k), ptr<int32_t>(ids),
                             ptr<float>(scores), ptr<int32_t>(count), stream_of(ids)), "rtrec_slim_similar_topk");
}

// ---- round 5: the device entry points that used to be called by raw ctypes, same typed checks ----
void store_decay_device(const at::Tensor &raw, const at::Tensor &ts, double rate, double now, at::Tensor out, at::Tensor unsafe_idx,
                        at::Tensor unsafe_count) {
    TORCH_CHECK(ts.numel() == raw.numel() && out.numel() == raw.numel(), "store_decay_device: raw, ts and out must have one length");
    check(abi().store_decay_device(ptr<const double>(raw), ptr<const double>(ts), raw.numel(), rate, now, ptr<float>(out),
                                   ptr<int32_t>(unsafe_idx), ptr<int32_t>(unsafe_count), static_cast<int32_t>(unsafe_idx.numel()),
                                   stream_of(out)), "rtrec_store_decay_device");
}

void store_fold_device(const at::Tensor &order, const at::Tensor &start, const at::Tensor &delta, const at::Tensor &tstamp, const OT &old,
                       double lo, double hi, bool upsert, at::Tensor out_val, at::Tensor out_ts, at::Tensor out_val32) {
    const int64_t g = start.numel() - 1;
    TORCH_CHECK(g >= 0 && out_val.numel() == g && out_ts.numel() == g && out_val32.numel() == g, "store_fold_device: outputs must hold one entry per group");
    TORCH_CHECK(delta.numel() == order.numel() && tstamp.numel() == order.numel(), "store_fold_device: order, delta and tstamp must have one length");
    check(abi().store_fold_device(ptr<const int64_t>(order), ptr<const int64_t>(start), g, ptr<const double>(delta), ptr<const double>(tstamp),
                                  ptr<const double>(old), lo, hi, upsert ? 1 : 0, ptr<double>(out_val), ptr<double>(out_ts),
                                  ptr<float>(out_val32), stream_of(out_val)), "rtrec_store_fold_device");
}

void fit_sgd_epochs(const at::Tensor &cptr, const at::Tensor &ttime, const at::Tensor &tval, const at::Tensor &targets, const at::Tensor &sel,
                    const at::Tensor &sel_count, int64_t n_users, int64_t n_items, int64_t nnz, int64_t cap, int64_t first_epoch,
                    int64_t n_epochs, int64_t max_iter, double tol, const at::Tensor &eta, const at::Tensor &ws_before,
                    const at::Tensor &ws_after, const at::Tensor &u_after, const at::Tensor &reset_cnt, const at::Tensor &reset_mult,
                    at::Tensor w, at::Tensor q, at::Tensor best_loss, at::Tensor no_improve, at::Tensor n_iter, at::Tensor unfinished) {
    check(abi().fit_sgd_epochs(static_cast<int32_t>(n_users), static_cast<int32_t>(n_items), ptr<const int32_t>(cptr), ptr<const int32_t>(ttime),
                               ptr<const float>(tval), nnz, ptr<const int32_t>(targets), static_cast<int32_t>(targets.numel()),
                               ptr<const int32_t>(sel), ptr<const int32_t>(sel_count), static_cast<int32_t>(cap),
                               static_cast<int32_t>(first_epoch), static_cast<int32_t>(n_epochs), static_cast<int32_t>(max_iter), tol,
                               ptr<const double>(eta), ptr<const double>(ws_before), ptr<const double>(ws_after), ptr<const double>(u_after),
                               ptr<const int32_t>(reset_cnt), ptr<const float>(reset_mult), ptr<float>(w), ptr<float>(q),
                               ptr<double>(best_loss), ptr<int32_t>(no_improve), ptr<int32_t>(n_iter), ptr<int32_t>(unfinished),
                               stream_of(w)), "rtrec_slim_fit_sgd_epochs");
}

void first_touch_aux(const OT &row_ids, const at::Tensor &xb_ptr, const at::Tensor &xb_col, int64_t n_rows, int64_t n_items,
                     const at::Tensor &wc_ptr, const at::Tensor &wc_row, int64_t top_k, const at::Tensor &ids, const at::Tensor &count,
                     at::Tensor aux) {
    check(abi().first_touch_aux(static_cast<int32_t>(n_rows), ptr<const int32_t>(row_ids), ptr<const int32_t>(xb_ptr), ptr<const int32_t>(xb_col),
                                static_cast<int32_t>(xb_ptr.size(0)) - 1, static_cast<int32_t>(n_items), ptr<const int32_t>(wc_ptr),
                                ptr<const int32_t>(wc_row), static_cast<int32_t>(top_k), ptr<const int32_t>(ids), ptr<const int32_t>(count),
                                ptr<uint32_t>(aux), stream_of(aux)), "rtrec_slim_first_touch_aux");
}

void dense_fill(const OT &row_ids, const at::Tensor &xb_ptr, const at::Tensor &xb_col, int64_t n_rows, int64_t col_lo, int64_t col_hi,
                int64_t top_k, bool filter_interacted, at::Tensor ids, at::Tensor scores, at::Tensor aux, at::Tensor count,
                const at::Tensor &flagged_in, at::Tensor flagged_out) {
    check(abi().dense_fill(static_cast<int32_t>(n_rows), ptr<const int32_t>(row_ids), ptr<const int32_t>(xb_ptr), ptr<const int32_t>(xb_col),
                           static_cast<int32_t>(xb_ptr.size(0)) - 1, static_cast<int32_t>(col_lo), static_cast<int32_t>(col_hi),
                           static_cast<int32_t>(top_k), filter_interacted ? 1 : 0, ptr<int32_t>(ids), ptr<float>(scores), ptr<uint32_t>(aux),
                           ptr<int32_t>(count), ptr<const int32_t>(flagged_in), ptr<int32_t>(flagged_out), stream_of(ids)),
          "rtrec_slim_dense_fill");
}

void refine_topk_f64(const OT &row_ids, const at::Tensor &xb_ptr, const at::Tensor &xb_col, const at::Tensor &xb_val, int64_t n_rows,
                     int64_t n_items, const at::Tensor &wc_ptr, const at::Tensor &wc_row, const at::Tensor &wc_val, int64_t top_k,
                     const at::Tensor &in_ids, const at::Tensor &in_scores, const at::Tensor &in_count, double rel_margin,
                     const OT &abs_slack, at::Tensor out_ids, at::Tensor out_scores, at::Tensor out_scores64, at::Tensor out_count,
                     at::Tensor flagged) {
    check(abi().refine_topk_f64(static_cast<int32_t>(n_rows), ptr<const int32_t>(row_ids), ptr<const int32_t>(xb_ptr), ptr<const int32_t>(xb_col),
                                ptr<const float>(xb_val), static_cast<int32_t>(xb_ptr.size(0)) - 1, static_cast<int32_t>(n_items),
                                ptr<const int32_t>(wc_ptr), ptr<const int32_t>(wc_row), ptr<const float>(wc_val), static_cast<int32_t>(top_k),
                                ptr<const int32_t>(in_ids), ptr<const float>(in_scores), ptr<const int32_t>(in_count), rel_margin,
                                ptr<const double>(abs_slack), ptr<int32_t>(out_ids), ptr<float>(out_scores), ptr<double>(out_scores64),
                                ptr<int32_t>(out_count), ptr<int32_t>(flagged), stream_of(out_ids)), "rtrec_slim_refine_topk_f64");
}

void score_candidates(const OT &row_ids, const at::Tensor &xb_ptr, const at::Tensor &xb_col, const at::Tensor &xb_val, int64_t n_rows,
                      int64_t n_items, const at::Tensor &wc_ptr, const at::Tensor &wc_row, const at::Tensor &wc_val, const at::Tensor &cands,
                      int64_t top_k, bool acc_f64, at::Tensor ids, at::Tensor scores, OT scores64, at::Tensor count) {
    check(abi().score_candidates(static_cast<int32_t>(n_rows), ptr<const int32_t>(row_ids), ptr<const int32_t>(xb_ptr), ptr<const int32_t>(xb_col),
                                 ptr<const float>(xb_val), static_cast<int32_t>(xb_ptr.size(0)) - 1, static_cast<int32_t>(n_items),
                                 ptr<const int32_t>(wc_ptr), ptr<const int32_t>(wc_row), ptr<const float>(wc_val), ptr<const int32_t>(cands),
                                 static_cast<int32_t>(cands.numel()), static_cast<int32_t>(top_k), acc_f64 ? 1 : 0, ptr<int32_t>(ids),
                                 ptr<float>(scores), ptr<double>(scores64), ptr<int32_t>(count), stream_of(ids)), "rtrec_slim_score_candidates");
}

// returns {n_cols, n_rows, tile_cols, n_tiles} of the segment layout (the C-ABI writes them to host memory after its own sync)
std::vector<int64_t> seg_plan(const at::Tensor &rows, const at::Tensor &cols, int64_t n_items, int64_t col_lo, int64_t col_hi,
                              const at::Tensor &labels, at::Tensor ws) {
    TORCH_CHECK(cols.numel() == rows.numel(), "seg_plan: rows and cols must have one length");
    int32_t out[4] = {0, 0, 0, 0};
    check(abi().seg_plan(static_cast<int32_t>(n_items), rows.numel(), ptr<const int64_t>(rows), ptr<const int64_t>(cols),
                         static_cast<int32_t>(col_lo), static_cast<int32_t>(col_hi), ptr<const int64_t>(labels), ptr(ws),
                         static_cast<size_t>(ws.numel()), out, stream_of(ws)), "rtrec_slim_seg_plan");
    return {out[0], out[1], out[2], out[3]};
}

void seg_fill(const at::Tensor &rows, const at::Tensor &cols, const at::Tensor &vals, int64_t n_items, int64_t col_lo, int64_t col_hi,
              const at::Tensor &plan_ws, int64_t n_cols, int64_t n_rows, int64_t tile_cols, int64_t n_tiles, at::Tensor ws, at::Tensor info,
              at::Tensor seg_ptr, at::Tensor ent, at::Tensor bound, at::Tensor col_ids, at::Tensor trow_ptr, at::Tensor trow) {
    TORCH_CHECK(cols.numel() == rows.numel() && vals.numel() == rows.numel(), "seg_fill: rows, cols and vals must have one length");
    check(abi().seg_fill(static_cast<int32_t>(n_items), rows.numel(), ptr<const int64_t>(rows), ptr<const int64_t>(cols), ptr<const float>(vals),
                         static_cast<int32_t>(col_lo), static_cast<int32_t>(col_hi), ptr<const void>(plan_ws), static_cast<int32_t>(n_cols),
                         static_cast<int32_t>(n_rows), static_cast<int32_t>(tile_cols), static_cast<int32_t>(n_tiles), ptr(ws),
                         static_cast<size_t>(ws.numel()), ptr<int32_t>(info), ptr<int32_t>(seg_ptr), ptr<int32_t>(ent), ent.size(0),
                         ptr<uint32_t>(bound), ptr<int32_t>(col_ids), ptr<int32_t>(trow_ptr), ptr<int32_t>(trow), trow.size(0),
                         stream_of(info)), "rtrec_slim_seg_fill");
}

void ordered_sums(const at::Tensor &values, const at::Tensor &offsets, int64_t mode, at::Tensor out) {
    TORCH_CHECK(offsets.numel() == out.numel() + 1, "ordered_sums: offsets must hold one more entry than out");
    check(abi().ordered_sums(ptr<const float>(values), ptr<const int64_t>(offsets), static_cast<int32_t>(out.numel()),
                             static_cast<int32_t>(mode), ptr<float>(out), stream_of(out)), "rtrec_slim_ordered_sums");
}

}  // namespace

// Bind the ops to a build of the C-ABI library (called once by rtrec_amd.ops with _native.lib_path()).
extern "C" int rtrec_ops_bind(const char *path) {
    try {
        void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return -1;
        Abi a;
        a.handle = h;
        bind_one(h, a.last_error, "rtrec_amd_last_error");
        bind_one(h, a.column_sqnorms, "rtrec_slim_column_sqnorms");
        bind_one(h, a.fit_workspace_init, "rtrec_slim_fit_workspace_init");
        bind_one(h, a.gram_matrix, "rtrec_slim_gram_matrix");
        bind_one(h, a.fit_columns_opt, "rtrec_slim_fit_columns_opt");
        bind_one(h, a.score_topk_opt, "rtrec_slim_score_topk_opt");
        bind_one(h, a.score_rows, "rtrec_slim_score_rows");
        bind_one(h, a.merge_topk_strided, "rtrec_slim_merge_topk_strided");
        bind_one(h, a.similar_topk, "rtrec_slim_similar_topk");
        bind_one(h, a.store_decay_device, "rtrec_store_decay_device");
        bind_one(h, a.store_fold_device, "rtrec_store_fold_device");
        bind_one(h, a.fit_sgd_epochs, "rtrec_slim_fit_sgd_epochs");
        bind_one(h, a.first_touch_aux, "rtrec_slim_first_touch_aux");
        bind_one(h, a.dense_fill, "rtrec_slim_dense_fill");
        bind_one(h, a.refine_topk_f64, "rtrec_slim_refine_topk_f64");
        bind_one(h, a.score_candidates, "rtrec_slim_score_candidates");
        bind_one(h, a.seg_plan, "rtrec_slim_seg_plan");
        bind_one(h, a.seg_fill, "rtrec_slim_seg_fill");
        bind_one(h, a.ordered_sums, "rtrec_slim_ordered_sums");
        g_abi = a;
        return 0;
    } catch (const std::exception &) {
        return -2;
    }
}

TORCH_LIBRARY(rtrec_amd, m) {
    m.def("column_sqnorms(Tensor cptr, Tensor cval, Tensor(a!) out) -> ()");
    m.def("fit_workspace_init(Tensor(a!) ws, int n_users, int n_items, int n_slots, int top_features) -> ()");
    m.def("gram_matrix(Tensor cptr, Tensor crow, Tensor cval, Tensor top_items, Tensor(a!) ws, Tensor(b!) gram, int n_users, int n_items) -> ()");
    m.def("fit_columns(Tensor cptr, Tensor crow, Tensor cval, Tensor rptr, Tensor rcol, Tensor rval, Tensor sqn, Tensor targets, "
          "int n_users, int n_items, float l1_reg, float l2_reg, float tol, int max_iter, int seed, bool positive, int top_features, "
          "Tensor(a!) out_items, Tensor(b!) out_coef, Tensor(c!) out_count, Tensor(d!) out_n_iter, int cap, Tensor(e!) ws, int n_slots, "
          "Tensor(f!) queue, Tensor(g!)? trace, Tensor? gram, Tensor? gram_index, int gram_n, float gram_rel_err, int fast, int kernel, "
          "int colwalk_min_rows, int screen_min, int lane_max, Tensor(h!)? xty_ws, Tensor? col_order, int fold) -> ()");
    m.def("score_topk(Tensor? row_ids, Tensor xb_ptr, Tensor xb_col, Tensor xb_val, int n_rows, int n_items, int n_cols, int col_offset, "
          "Tensor? col_ids, Tensor? col_map, int tile_cols, int n_tiles, Tensor? tile_ptr, Tensor? w_col, Tensor? w_val, Tensor? dense_idx, "
          "Tensor? dense_val, Tensor? row_hdr, Tensor? col_rank, int top_k, bool filter_interacted, int mode, bool acc_f64, "
          "Tensor(a!) ids, Tensor(b!) scores, Tensor(c!)? scores64, Tensor(d!) aux, Tensor(e!) count, Tensor(f!) ws, Tensor? fr_map, "
          "Tensor? fr_col_ids, Tensor? fr_col_map, Tensor? fr_w, Tensor? fr_tile_rows, Tensor? fr_tile_off, Tensor? fr_super_kb, "
          "Tensor? fr_super_tile, Tensor? fr_frag_tile, int fr_rows, int fr_tile_cols, int fr_n_tiles, int fr_n_frags, int fr_n_super, "
          "int fr_buf_bytes, Tensor(g!)? fr_scratch, Tensor? row_order, int timer, int diagnostics, Tensor(h!)? rescored, "
          "int row_order_grouped, Tensor? sg_info, Tensor? sg_ptr, Tensor? sg_ent, Tensor? sg_bound, Tensor? sg_col_ids, int sg_tile_cols, "
          "int sg_n_tiles, int sg_rows, int sg_n_cols, Tensor? sg_trow_ptr, Tensor? sg_trow, Tensor(i!)? sg_scratch, "
          "Tensor(j!)? flagged, int aux_stream) -> ()");
    m.def("score_rows(Tensor? row_ids, Tensor xb_ptr, Tensor xb_col, Tensor xb_val, int n_rows, int n_items, int n_cols, int col_offset, "
          "int tile_cols, int n_tiles, Tensor tile_ptr, Tensor w_col, Tensor w_val, bool acc_f64, Tensor(a!) out) -> ()");
    m.def("merge_topk(Tensor in_ids, Tensor in_scores, Tensor? in_scores64, Tensor in_aux, Tensor in_count, int top_k, "
          "Tensor(a!) out_ids, Tensor(b!) out_scores, Tensor(c!) out_count) -> ()");
    m.def("similar_topk(Tensor queries, Tensor wc_ptr, Tensor wc_row, Tensor wc_val, int top_k, Tensor(a!) ids, Tensor(b!) scores, "
          "Tensor(c!) count) -> ()");
    m.def("store_decay_device(Tensor raw, Tensor ts, float rate, float now, Tensor(a!) out, Tensor(b!) unsafe_idx, Tensor(c!) unsafe_count) -> ()");
    m.def("store_fold_device(Tensor order, Tensor start, Tensor delta, Tensor tstamp, Tensor? old, float lo, float hi, bool upsert, "
          "Tensor(a!) out_val, Tensor(b!) out_ts, Tensor(c!) out_val32) -> ()");
    m.def("fit_sgd_epochs(Tensor cptr, Tensor ttime, Tensor tval, Tensor targets, Tensor sel, Tensor sel_count, int n_users, int n_items, "
          "int nnz, int cap, int first_epoch, int n_epochs, int max_iter, float tol, Tensor eta, Tensor ws_before, Tensor ws_after, "
          "Tensor u_after, Tensor reset_cnt, Tensor reset_mult, Tensor(a!) w, Tensor(b!) q, Tensor(c!) best_loss, Tensor(d!) no_improve, "
          "Tensor(e!) n_iter, Tensor(f!) unfinished) -> ()");
    m.def("first_touch_aux(Tensor? row_ids, Tensor xb_ptr, Tensor xb_col, int n_rows, int n_items, Tensor wc_ptr, Tensor wc_row, int top_k, "
          "Tensor ids, Tensor count, Tensor(a!) aux) -> ()");
    m.def("dense_fill(Tensor? row_ids, Tensor xb_ptr, Tensor xb_col, int n_rows, int col_lo, int col_hi, int top_k, bool filter_interacted, "
          "Tensor(a!) ids, Tensor(b!) scores, Tensor(c!) aux, Tensor(d!) count, Tensor flagged_in, Tensor(e!) flagged_out) -> ()");
    m.def("refine_topk_f64(Tensor? row_ids, Tensor xb_ptr, Tensor xb_col, Tensor xb_val, int n_rows, int n_items, Tensor wc_ptr, "
          "Tensor wc_row, Tensor wc_val, int top_k, Tensor in_ids, Tensor in_scores, Tensor in_count, float rel_margin, Tensor? abs_slack, "
          "Tensor(a!) out_ids, Tensor(b!) out_scores, Tensor(c!) out_scores64, Tensor(d!) out_count, Tensor(e!) flagged) -> ()");
    m.def("score_candidates(Tensor? row_ids, Tensor xb_ptr, Tensor xb_col, Tensor xb_val, int n_rows, int n_items, Tensor wc_ptr, "
          "Tensor wc_row, Tensor wc_val, Tensor cands, int top_k, bool acc_f64, Tensor(a!) ids, Tensor(b!) scores, Tensor(c!)? scores64, "
          "Tensor(d!) count) -> ()");
    m.def("seg_plan(Tensor rows, Tensor cols, int n_items, int col_lo, int col_hi, Tensor labels, Tensor(a!) ws) -> int[]");
    m.def("seg_fill(Tensor rows, Tensor cols, Tensor vals, int n_items, int col_lo, int col_hi, Tensor plan_ws, int n_cols, int n_rows, "
          "int tile_cols, int n_tiles, Tensor(a!) ws, Tensor(b!) info, Tensor(c!) seg_ptr, Tensor(d!) ent, Tensor(e!) bound, "
          "Tensor(f!) col_ids, Tensor(g!) trow_ptr, Tensor(h!) trow) -> ()");
    m.def("ordered_sums(Tensor values, Tensor offsets, int mode, Tensor(a!) out) -> ()");
}

TORCH_LIBRARY_IMPL(rtrec_amd, CUDA, m) {
    m.impl("column_sqnorms", &column_sqnorms);
    m.impl("fit_workspace_init", &fit_workspace_init);
    m.impl("gram_matrix", &gram_matrix);
    m.impl("fit_columns", &fit_columns);
    m.impl("score_topk", &score_topk);
    m.impl("score_rows", &score_rows);
    m.impl("merge_topk", &merge_topk);
    m.impl("similar_topk", &similar_topk);
    m.impl("store_decay_device", &store_decay_device);
    m.impl("store_fold_device", &store_fold_device);
    m.impl("fit_sgd_epochs", &fit_sgd_epochs);
    m.impl("first_touch_aux", &first_touch_aux);
    m.impl("dense_fill", &dense_fill);
    m.impl("refine_topk_f64", &refine_topk_f64);
    m.impl("score_candidates", &score_candidates);
    m.impl("seg_plan", &seg_plan);
    m.impl("seg_fill", &seg_fill);
    m.impl("ordered_sums", &ordered_sums);
}
