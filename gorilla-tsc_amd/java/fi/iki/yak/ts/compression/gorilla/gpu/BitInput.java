package fi.iki.yak.ts.compression.gorilla.gpu;

/** What a decompressor reads: a word image (the role of G/BitInput.java:8-32). */
public abstract class BitInput {
    abstract long[] words();
}
