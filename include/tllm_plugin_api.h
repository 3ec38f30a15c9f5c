/*
 * tllm_plugin_api.h — the drop-in boundary of the MI355X LLaMA decoder hot path.
 *
 * A flat C ABI that replaces, entry point for entry point, what the reference binds through
 * TensorRT's C++ plugin vtables (TensorRT does not exist on MI355X).  Path shorthands follow
 * T/ = tensorrt_llm_july-release-v1/, P/ = T/cpp/tensorrt_llm/plugins/,
 * PY/ = T/tensorrt_llm/.
 *
 * Conventions
 *   - every tensor pointer is a DEVICE pointer unless stated otherwise; the caller owns all
 *     input / output / workspace buffers (P/gptAttentionPlugin/gptAttentionPlugin.cpp:132-144);
 *   - work is enqueued on the hipStream_t that is passed in, no hidden synchronisation;
 *   - functions returning int32_t return 0 on success and non-zero on failure; the reason is
 *     available (thread-local) from tllm_last_error(); nothing here ever calls exit()/abort()
 *     (the reference does: P/common/plugin.h:159-197);
 *   - data type codes are nvinfer1::DataType's, tensor descriptors are nvinfer1::PluginTensorDesc's
 *     field for field, plugin fields are nvinfer1::PluginField's.
 */
#ifndef TLLM_PLUGIN_API_H
#define TLLM_PLUGIN_API_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without dragging the HIP headers into a C consumer. */
typedef struct ihipStream_t* tllm_stream_t;

/* nvinfer1::DataType */
enum
{
    TLLM_FLOAT = 0,
    TLLM_HALF = 1,
    TLLM_INT8 = 2,
    TLLM_INT32 = 3,
    TLLM_BOOL = 4,
    TLLM_UINT8 = 5,
    TLLM_FP8 = 6
};

/* nvinfer1::PluginFieldType */
enum
{
    TLLM_FIELD_FLOAT16 = 0,
    TLLM_FIELD_FLOAT32 = 1,
    TLLM_FIELD_FLOAT64 = 2,
    TLLM_FIELD_INT8 = 3,
    TLLM_FIELD_INT16 = 4,
    TLLM_FIELD_INT32 = 5,
    TLLM_FIELD_CHAR = 6,
    TLLM_FIELD_DIMS = 7,
    TLLM_FIELD_UNKNOWN = 8
};

#define TLLM_MAX_DIMS 8

/* nvinfer1::Dims */
typedef struct
{
    int32_t nbDims;
    int32_t d[TLLM_MAX_DIMS];
} tllm_dims_t;

/* nvinfer1::PluginTensorDesc {dims, type, format(kLINEAR=0), scale} */
typedef struct
{
    tllm_dims_t dims;
    int32_t type;
    int32_t format;
    float scale;
} tllm_tensor_desc_t;

/* nvinfer1::PluginField {name, data, type, length} */
typedef struct
{
    const char* name;
    const void* data;
    int32_t type;
    int32_t length;
} tllm_plugin_field_t;

typedef struct tllm_plugin* tllm_plugin_t;

/* ------------------------------------------------------------------------------------------------
 * Library init.  Replaces  bool initLibNvInferPlugins(void* logger, const char* libNamespace)
 * (P/api/InferPlugin.cpp:149-171), which PY/plugin/plugin.py:7-22 calls through ctypes and
 * asserts to be true.  Idempotent and thread-safe like the mutex-guarded registry
 * (P/api/InferPlugin.cpp:55-136).  `logger` may be NULL.  Same signature as the reference's (bool: the reference's
 * ctypes stub declares restype c_bool and needs no edit).
 * ---------------------------------------------------------------------------------------------- */
bool initLibNvInferPlugins(void* logger, const char* libNamespace);

/* getInferLibVersion of P/exports.map:19-32 — here: 10000*major + 100*minor + patch of this library. */
int32_t getInferLibVersion(void);

/* Thread-local reason for the last failure on this thread ("" if none). */
const char* tllm_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Plugin registry.  Replaces
 *   trt.get_plugin_registry().get_plugin_creator(name, '1', 'tensorrt_llm')  and
 *   IPluginCreator::createPlugin(name, PluginFieldCollection*)
 * (PY/functional.py:2826-2893, PY/quantization/functional.py:12-212, PY/layers/linear.py:13-35).
 * Registered names (P/api/InferPlugin.cpp:153-168 subset):
 *   "GPTAttention", "Gemm", "SmoothQuantGemm", "WeightOnlyQuantMatmul", "QuantizeTensor",
 *   "QuantizePerToken", "LayernormQuantization", "AllReduce", "AllGather"
 * plus the MI355X additions that the reference composes out of TensorRT pointwise layers:
 *   "Rmsnorm", "RmsnormQuantization", "SwiGLU"
 * Field names / types / defaults are the reference's; an unknown or missing field makes creation
 * fail and returns NULL (reference: std::optional::value() throws, caught, nullptr —
 * P/gptAttentionPlugin/gptAttentionPlugin.cpp:483-511).
 * ---------------------------------------------------------------------------------------------- */
int32_t tllm_plugin_registry_size(void);
const char* tllm_plugin_registry_name(int32_t index);

tllm_plugin_t tllm_plugin_create(const char* name, const char* version, const char* ns,
    const tllm_plugin_field_t* fields, int32_t nbFields);

/* IPluginV2::getPluginType / getPluginVersion / getNbOutputs */
const char* tllm_plugin_type(tllm_plugin_t p);
const char* tllm_plugin_version(tllm_plugin_t p);
int32_t tllm_plugin_nb_outputs(tllm_plugin_t p);

/* IPluginV2DynamicExt::getOutputDimensions, evaluated on concrete input dims. */
int32_t tllm_plugin_output_dims(
    tllm_plugin_t p, int32_t outputIndex, const tllm_dims_t* inputs, int32_t nbInputs, tllm_dims_t* out);

/* IPluginV2Ext::getOutputDataType */
int32_t tllm_plugin_output_dtype(tllm_plugin_t p, int32_t outputIndex, const int32_t* inputTypes, int32_t nbInputs);

/* IPluginV2DynamicExt::supportsFormatCombination */
int32_t tllm_plugin_supports_format(
    tllm_plugin_t p, int32_t pos, const tllm_tensor_desc_t* inOut, int32_t nbInputs, int32_t nbOutputs);

/* IPluginV2DynamicExt::getWorkspaceSize (GPTAttention: max(context, generation),
 * P/gptAttentionPlugin/gptAttentionPlugin.cpp:132-144). */
size_t tllm_plugin_workspace_size(tllm_plugin_t p, const tllm_tensor_desc_t* inputs, int32_t nbInputs,
    const tllm_tensor_desc_t* outputs, int32_t nbOutputs);

/* IPluginV2DynamicExt::enqueue(inputDesc, outputDesc, inputs, outputs, workspace, stream) -> 0 on success.
 * Input order per plugin is the reference's.  All pointers are device pointers
 * except GPTAttention input 3 (past_key_value_length, host int32[2], PY/runtime/generation.py:579,689). */
int32_t tllm_plugin_enqueue(tllm_plugin_t p, const tllm_tensor_desc_t* inputDesc, const tllm_tensor_desc_t* outputDesc,
    const void* const* inputs, void* const* outputs, void* workspace, tllm_stream_t stream);

/* IPluginV2::getSerializationSize / serialize, IPluginCreator::deserializePlugin, IPluginV2DynamicExt::clone,
 * IPluginV2::destroy.  The byte format is this library's own (P/common/plugin.h:90-101 is a raw POD memcpy
 * with no cross-version guarantee either). */
size_t tllm_plugin_serialization_size(tllm_plugin_t p);
int32_t tllm_plugin_serialize(tllm_plugin_t p, void* buffer);
tllm_plugin_t tllm_plugin_deserialize(const char* name, const void* data, size_t length);
tllm_plugin_t tllm_plugin_clone(tllm_plugin_t p);
void tllm_plugin_destroy(tllm_plugin_t p);

/* ------------------------------------------------------------------------------------------------
 * Tensor-parallel communicator (RCCL over xGMI).  Replaces the MPI bootstrap of
 * P/ncclPlugin/allreducePlugin.cpp:124-162: the host front-end (one process per GPU) obtains the
 * 128-byte unique id on the lowest rank of the group, ships it to the peers by whatever transport it
 * has (torch.distributed store here, MPI_Send/Recv there) and every rank registers the communicator
 * for its rank set.  AllReduce / AllGather plugins look it up by their `group` field.
 * ---------------------------------------------------------------------------------------------- */
#define TLLM_COMM_ID_BYTES 128
int32_t tllm_comm_get_unique_id(void* id128);
int32_t tllm_comm_init_rank(const int32_t* group, int32_t groupSize, int32_t rank, const void* id128);
int32_t tllm_comm_destroy_all(void);

/* One-shot peer-to-peer all-reduce (fp16 sum, in place) for the 8 KB partial sums of the tensor-parallel decode step:
 * every rank writes its vector into a slot of every peer's inbox over xGMI, raises a flag, and adds the slots up in rank
 * order (bit-identical on all ranks) - one hop instead of a ring (no reference counterpart, the
 * reference calls ncclAllReduce: P/ncclPlugin/allreducePlugin.cpp:93).
 *   create : allocates this rank's inbox (uncached device memory, 2 x world x max_bytes) and returns its 64-byte
 *            hipIpcMemHandle_t in `handle64`;
 *   attach : `handles` = the world x 64 bytes of every rank's handle in rank order (exchanged by the caller, like the
 *            unique id);
 *   enable : lets the AllReduce plugin and the session use it for fp16 vectors of at most max_bytes (after the caller
 *            has validated it against the RCCL result on every rank; default off = RCCL);
 *   all_reduce : direct entry (tests); spins are bounded, tllm_comm_p2p_error() returns non-zero after a time-out. */
int32_t tllm_comm_p2p_create(int32_t world, int32_t rank, int64_t max_bytes, void* handle64);
int32_t tllm_comm_p2p_attach(const void* handles);
int32_t tllm_comm_p2p_enable(int32_t on); /* non-zero: refused (not attached; or out of service after a time-out - the ranks'
                                             epochs are no longer in step - until create + attach on every rank) */
/* The verdict of the caller's validation of the fused layer seam below (default on).  Off keeps the all-reduce peer-to-peer
 * and the residual add / RMSNorm / quantiser in the consuming kernels.  Sessions pick the change up at their next step (a
 * captured step graph is re-captured). */
void tllm_comm_p2p_enable_fused(int32_t on);
/* bit 0 attached, bit 1 enabled (the decode all-reduce runs peer-to-peer), bit 2 the fused layer seam is in use */
int32_t tllm_comm_p2p_state(void);
int32_t tllm_comm_p2p_all_reduce(void* buf, int64_t count, tllm_stream_t stream);
int32_t tllm_comm_p2p_error(void);
/* The tensor-parallel layer seam in ONE launch: all-reduce + residual add + the next RMSNorm (+ SmoothQuant activation quantiser).
 * Reference seam, three to four graph nodes per rank: allreduce plugin (P/ncclPlugin/allreducePlugin.cpp:80-96, inserted by
 * PY/quantization/layer.py:215,377 / PY/layers/linear.py:127-139) -> `hidden = residual + ...` (Q/llama_model.py:107-108,
 * :117-118) -> rms_norm (PY/functional.py:3195-3219) [-> quantize, PY/quantization/layer.py:223-265].
 *   partial  : this rank's [rows, cols] fp16 partial sum (read only);
 *   x        : [rows, cols] fp16 residual stream, in place: x <- fp16(x + sum_r partial_r)  (fp32 sum in rank order, one rounding,
 *              bit-identical on every rank);
 *   norm_out : [rows, cols]  fp16(fp16(x * rsqrt(mean(x^2) + eps)) * gamma)          (quant 0)
 *              int8 sat(rni(that * quant_scale[0]))                                   (quant 1, static)
 *              int8 per token, amax / 127 to dyn_scale_out[rows]                      (quant 2; K/quantization.cu:94-118).
 * rows * cols * 2 bytes must fit the inbox slot (tllm_comm_p2p_create max_bytes). */
int32_t tllm_comm_p2p_all_reduce_residual_norm(const void* partial, void* x, const void* gamma, float eps, int32_t rows, int32_t cols,
    void* norm_out, int32_t quant, const float* quant_scale, float* dyn_scale_out, tllm_stream_t stream);
/* Test knob: how many polls a flag wait may take before it gives up (0 = default, about a second). */
void tllm_comm_p2p_set_max_spins(int32_t n);
/* What the RCCL communicator registered for `group` reports about itself (ncclCommCount / ncclCommUserRank): the number of ranks
 * it spans and this process's index in it; returns non-zero when no communicator exists for the group. */
int32_t tllm_comm_group_info(const int32_t* group, int32_t groupSize, int32_t* nranks, int32_t* my_index);

/* ------------------------------------------------------------------------------------------------
 * Weight pre-processing for the weight-only plugins.  Replaces the torch ops of
 * T/cpp/tensorrt_llm/thop/weightOnlyQuantOp.cpp:143-236,343-371 that the loaders call
 * (T/examples/llama/weight.py:101-110, T/examples/llama_quant/weight_quant.py:264):
 *   symmetric_quantize_last_axis_of_batched_matrix(W[k,n] fp16, quant_type) ->
 *        (processed int8 weights for the GEMM, fp16 scales[n])
 * Host (CPU) function; `weight_kn` is row-major [k, n] fp16 bits.  bits = 8 or 4.
 * processed layout (this library's own, replaces the SM80 interleave of
 * T/cpp/tensorrt_llm/kernels/cutlass_kernels/cutlass_preprocessors.cpp:158-535):
 *   int8: [n, roundup(k,16)] row-major (k contiguous), stored as u8 = q + 128;
 *   int4: [n, roundup(k,32)/2], nibble = q + 8, the 8 nibbles of every 32-bit word (low to high) hold elements
 *         e0 e2 e4 e6 e1 e3 e5 e7 of that group of 8.
 * `unprocessed_out` (optional, may be NULL) receives the plain quantised [k, n] (int8) or [k, n/2] (int4,
 * low nibble first) tensor — the op's 3-output variant `_symmetric_quantize_last_axis_of_batched_matrix`.
 * ---------------------------------------------------------------------------------------------- */
int32_t tllm_symmetric_quantize_last_axis(const uint16_t* weight_kn, int64_t k, int64_t n, int32_t bits,
    int8_t* processed_out, uint16_t* scales_out, int8_t* unprocessed_out);

/* preprocess_weights_for_mixed_gemm(int8 [k,n] or packed int4 [k,n/2]) -> processed layout above. */
int32_t tllm_preprocess_weights_for_mixed_gemm(
    const int8_t* quantized_kn, int64_t k, int64_t n, int32_t bits, int8_t* processed_out);

#ifdef __cplusplus
}
#endif

#endif /* TLLM_PLUGIN_API_H */
