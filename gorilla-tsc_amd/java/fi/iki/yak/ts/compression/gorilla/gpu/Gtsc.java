package fi.iki.yak.ts.compression.gorilla.gpu;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Downcall handles for libgtsc.so (include/gtsc.h) through java.lang.foreign (JDK 22+).
 * No JNI glue: the library's C ABI uses only pointers and 32/64-bit integers.
 *
 * Source only: there is no JDK in the image this library is built and tested in.
 */
public final class Gtsc {
    public static final int OK = 0, ERR_ARG = -1, ERR_CAPACITY = -2, ERR_HIP = -3, ERR_STREAM = -4;
    public static final int FMT_V2 = 0, FMT_V1 = 1;
    public static final int ST_OVERFLOW = 1, ST_TRUNCATED = 2, ST_MALFORMED = 4;
    public static final int PRED_LAST = 0, PRED_DFCM = 1;
    public static final int ENC_NO_CLOSE = 1;

    private static final Linker LINKER = Linker.nativeLinker();
    private static final SymbolLookup LIB =
            SymbolLookup.libraryLookup(System.getProperty("gtsc.library", "libgtsc.so"), Arena.global());

    private static MethodHandle h(String name, FunctionDescriptor fd) {
        return LINKER.downcallHandle(LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), fd);
    }

    static final MethodHandle VERSION = h("gtsc_version", FunctionDescriptor.of(JAVA_INT));
    static final MethodHandle LAST_ERROR = h("gtsc_last_error", FunctionDescriptor.of(ADDRESS));
    static final MethodHandle MAX_WORDS = h("gtsc_max_words", FunctionDescriptor.of(JAVA_LONG, JAVA_LONG));
    static final MethodHandle CREATE = h("gtsc_create", FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS));
    static final MethodHandle DESTROY = h("gtsc_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
    static final MethodHandle HOST_ALLOC = h("gtsc_host_alloc", FunctionDescriptor.of(ADDRESS, JAVA_LONG));
    static final MethodHandle HOST_FREE = h("gtsc_host_free", FunctionDescriptor.ofVoid(ADDRESS));
    /* int gtsc_encode_host(ctx, format, nseries, block_ts, ts, val, pt_off, uniform_npoints,
     *                      out_words, out_capacity_words, out_off) */
    static final MethodHandle ENCODE_HOST = h("gtsc_encode_host", FunctionDescriptor.of(JAVA_INT,
            ADDRESS, JAVA_INT, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, ADDRESS));
    /* int gtsc_encode_host_ex(ctx, format, options, nseries, block_ts, ts, val, pt_off, uniform_npoints,
     *                         out_words, out_capacity_words, out_off, out_nbits) */
    static final MethodHandle ENCODE_HOST_EX = h("gtsc_encode_host_ex", FunctionDescriptor.of(JAVA_INT,
            ADDRESS, JAVA_INT, JAVA_INT, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));
    /* int gtsc_encode_host_multi(nctx, ctxs, format, options, nseries, block_ts, ts, val, pt_off, uniform_npoints,
     *                            out_words, out_capacity_words, out_off, out_nbits) */
    static final MethodHandle ENCODE_HOST_MULTI = h("gtsc_encode_host_multi", FunctionDescriptor.of(JAVA_INT,
            JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));
    /* int gtsc_decode_host_multi(nctx, ctxs, format, nseries, words, in_off, ts, val, pt_off, uniform_capacity,
     *                            out_npoints, status) */
    static final MethodHandle DECODE_HOST_MULTI = h("gtsc_decode_host_multi", FunctionDescriptor.of(JAVA_INT,
            JAVA_INT, ADDRESS, JAVA_INT, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));
    /* int gtsc_set_tunable(ctx, which, value) */
    static final MethodHandle SET_TUNABLE = h("gtsc_set_tunable", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_LONG));
    /* int gtsc_set_predictor(ctx, predictor, table_size) */
    static final MethodHandle SET_PREDICTOR = h("gtsc_set_predictor", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT));
    /* int gtsc_decode_host(ctx, format, nseries, words, in_off, ts, val, pt_off, uniform_capacity,
     *                      out_npoints, status) */
    static final MethodHandle DECODE_HOST = h("gtsc_decode_host", FunctionDescriptor.of(JAVA_INT,
            ADDRESS, JAVA_INT, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));

    private Gtsc() { }

    static String lastError() {
        try {
            MemorySegment s = (MemorySegment) LAST_ERROR.invokeExact();
            return s.reinterpret(4096).getString(0);
        } catch (Throwable t) {
            return t.toString();
        }
    }

    static long maxWords(long npoints) {
        try {
            return (long) MAX_WORDS.invokeExact(npoints);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** A context on the given device (-1: the current one), e.g. one per GPU for a GpuBatch over the node. */
    public static MemorySegment createContext(int device) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment out = a.allocate(ADDRESS);
            int rc = (int) CREATE.invokeExact(device, out);
            if (rc != OK) throw new IllegalStateException(lastError());
            return out.get(ADDRESS, 0);
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** One gtsc_ctx per thread, like the reference's non-thread-safe per-instance objects. */
    static final ThreadLocal<MemorySegment> CONTEXT = ThreadLocal.withInitial(() -> createContext(-1));
}
