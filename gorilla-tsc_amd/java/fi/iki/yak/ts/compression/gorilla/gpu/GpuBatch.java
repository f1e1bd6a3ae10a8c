package fi.iki.yak.ts.compression.gorilla.gpu;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.ArrayList;
import java.util.List;

import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Collects closed blocks and codes them with one gtsc_encode_host call.  A foreign call per
 * point would cost more than the codec, so addValue() only buffers; flush() does the work.
 */
public final class GpuBatch {
    private final List<GorillaCompressor> closed = new ArrayList<>();

    void add(GorillaCompressor c) { closed.add(c); }

    /** Codes every closed block; each block's LongArrayOutput then holds its image. */
    public void flush() {
        for (int format : new int[] {Gtsc.FMT_V2, Gtsc.FMT_V1}) {
            List<GorillaCompressor> blocks = closed.stream().filter(c -> c.format == format).toList();
            if (blocks.isEmpty()) continue;
            int S = blocks.size();
            long total = 0, cap = 0;
            for (GorillaCompressor c : blocks) { total += c.size; cap += Gtsc.maxWords(c.size); }
            try (Arena a = Arena.ofConfined()) {
                MemorySegment blockTs = a.allocate(JAVA_LONG, S);
                MemorySegment ptOff = a.allocate(JAVA_LONG, S + 1L);
                MemorySegment ts = a.allocate(JAVA_LONG, Math.max(total, 1));
                MemorySegment val = a.allocate(JAVA_LONG, Math.max(total, 1));
                MemorySegment words = a.allocate(JAVA_LONG, cap);
                MemorySegment outOff = a.allocate(JAVA_LONG, S + 1L);
                long p = 0;
                for (int s = 0; s < S; s++) {
                    GorillaCompressor c = blocks.get(s);
                    blockTs.setAtIndex(JAVA_LONG, s, c.blockTimestamp);
                    ptOff.setAtIndex(JAVA_LONG, s, p);
                    MemorySegment.copy(c.ts, 0, ts, JAVA_LONG, p * 8, c.size);
                    MemorySegment.copy(c.val, 0, val, JAVA_LONG, p * 8, c.size);
                    p += c.size;
                }
                ptOff.setAtIndex(JAVA_LONG, S, p);
                int rc = (int) Gtsc.ENCODE_HOST.invokeExact(Gtsc.CONTEXT.get(), format, (long) S, blockTs, ts, val, ptOff, 0L,
                        words, cap, outOff);
                if (rc != Gtsc.OK) throw new IllegalStateException(Gtsc.lastError());
                for (int s = 0; s < S; s++) {
                    long b = outOff.getAtIndex(JAVA_LONG, s), e = outOff.getAtIndex(JAVA_LONG, s + 1L);
                    long[] image = new long[(int) (e - b)];
                    MemorySegment.copy(words, JAVA_LONG, b * 8, image, 0, image.length);
                    blocks.get(s).out.set(image);
                }
            } catch (RuntimeException e) {
                throw e;
            } catch (Throwable t) {
                throw new IllegalStateException(t);
            }
        }
        closed.clear();
    }

    /** Decodes one image; returns {timestamps, raw values} (the arrays' length is the pair count). */
    static long[][] decode(int format, long[] image) {
        long cap = image.length * 32L + 1;  // a pair takes at least two bits
        try (Arena a = Arena.ofConfined()) {
            MemorySegment words = a.allocate(JAVA_LONG, Math.max(image.length, 1));
            MemorySegment.copy(image, 0, words, JAVA_LONG, 0, image.length);
            MemorySegment inOff = a.allocate(JAVA_LONG, 2);
            inOff.setAtIndex(JAVA_LONG, 1, image.length);
            MemorySegment ts = a.allocate(JAVA_LONG, cap), val = a.allocate(JAVA_LONG, cap);
            MemorySegment n = a.allocate(JAVA_LONG), st = a.allocate(JAVA_INT);
            int rc = (int) Gtsc.DECODE_HOST.invokeExact(Gtsc.CONTEXT.get(), format, 1L, words, inOff, ts, val,
                    MemorySegment.NULL, cap, n, st);
            if (rc != Gtsc.OK && rc != Gtsc.ERR_STREAM) throw new IllegalStateException(Gtsc.lastError());
            int status = st.get(JAVA_INT, 0);
            if ((status & (Gtsc.ST_TRUNCATED | Gtsc.ST_MALFORMED)) != 0)
                throw new ArrayIndexOutOfBoundsException("compressed array ended inside a value");  // what LongArrayInput throws
            int count = (int) n.get(JAVA_LONG, 0);
            long[][] r = new long[2][count];
            MemorySegment.copy(ts, JAVA_LONG, 0, r[0], 0, count);
            MemorySegment.copy(val, JAVA_LONG, 0, r[1], 0, count);
            return r;
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
