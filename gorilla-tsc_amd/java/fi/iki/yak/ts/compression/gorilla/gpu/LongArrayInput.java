package fi.iki.yak.ts.compression.gorilla.gpu;

public class LongArrayInput extends BitInput {
    private final long[] array;

    public LongArrayInput(long[] array) { this.array = array; }

    @Override
    long[] words() { return array; }
}
